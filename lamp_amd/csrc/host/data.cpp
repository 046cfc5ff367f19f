// lamp-data's tensor-list files, checkpoints, the CIFAR record reader and the minibatch stream, over the C ABI.
//
// Reference (behaviour restated, nothing copied):
//   lamp-data/src/main/scala/lamp/data/Writer.scala:14-190      format spec + writeTensorsIntoFile / writeCheckpoint
//   lamp-data/src/main/scala/lamp/data/Reader.scala:17-95       readTensorsFromFile / loadFromFile
//   lamp-data/src/main/scala/lamp/data/schemas/schemas.scala:30-56   TensorDescriptor / TensorList (the JSON descriptor)
//   lamp-sten/src/main/scala/lamp/STen.scala:148-194            tensorsFromFile: alignment / bounds assertions, empty lists
//   example-cifar100/src/main/scala/lamp/example/cifar/cifar100.scala:29-56   3074-byte records
//   lamp-data/src/main/scala/lamp/data/BatchStream.scala:378-400 (everyNth), :528-592 (minibatchesFromFull)
//
// MI355X-first differences (same results):
//   * the reference gathers every minibatch on the host, stages it in a pinned buffer and copies it to the GPU
//     (Device.toBatched).  With 288 GB of HBM the whole data set is uploaded once and a minibatch is one index_select
//     kernel on the device: no per-step PCIe traffic, no host gather.
//   * the CIFAR file is uploaded as raw bytes; label / pixel split and the cast run on the GPU.
//   * files are read with plain reads into (optionally pinned) host memory, not mmap.
//   * a data set that must stay in host memory (lamp_batch_stream_from_full_host) is kept pinned; the GPU gathers a minibatch's rows over
//     PCIe itself (lamp_index_select_pinned), queued one batch ahead - no host gather, no staging buffer.
#include <sys/stat.h>

#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>

#include "../../../include/lamp_host.h"
#include "nn.h"

namespace lamp {
namespace host {
namespace {

struct TensorDescriptor { std::vector<int64_t> dims; int dataType = 0; int64_t byteOffset = 0, byteLength = 0; };
struct TensorList { std::vector<TensorDescriptor> tensors; std::string location; int64_t byteOffset = 0, byteLength = 0; };

// ---- the JSON subset of the descriptor: objects, arrays, strings, integers, null / true / false --------------
struct Json {
  const std::string& s; size_t i = 0;
  explicit Json(const std::string& src) : s(src) {}
  void ws() { while (i < s.size() && (s[i] == ' ' || s[i] == '\n' || s[i] == '\t' || s[i] == '\r')) i++; }
  char peek() { ws(); LAMP_CHECK(i < s.size(), "tensor list descriptor: unexpected end of JSON"); return s[i]; }
  void expect(char c) { LAMP_CHECK(peek() == c, "tensor list descriptor: expected '" << c << "' at byte " << i); i++; }
  bool accept(char c) { if (peek() == c) { i++; return true; } return false; }
  std::string str() {
    expect('"');
    std::string o;
    while (true) {
      LAMP_CHECK(i < s.size(), "tensor list descriptor: unterminated string");
      char c = s[i++];
      if (c == '"') break;
      if (c == '\\') {
        LAMP_CHECK(i < s.size(), "tensor list descriptor: bad escape");
        char e = s[i++];
        switch (e) {
          case 'n': o += '\n'; break; case 't': o += '\t'; break; case 'r': o += '\r'; break; case 'b': o += '\b'; break;
          case 'f': o += '\f'; break; case '/': o += '/'; break; case '\\': o += '\\'; break; case '"': o += '"'; break;
          case 'u': {
            LAMP_CHECK(i + 4 <= s.size(), "tensor list descriptor: bad \\u escape");
            unsigned cp = (unsigned)std::stoul(s.substr(i, 4), nullptr, 16); i += 4;
            if (cp < 0x80) o += (char)cp;
            else if (cp < 0x800) { o += (char)(0xC0 | (cp >> 6)); o += (char)(0x80 | (cp & 0x3F)); }
            else { o += (char)(0xE0 | (cp >> 12)); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
            break;
          }
          default: throw Error("tensor list descriptor: bad escape");
        }
      } else o += c;
    }
    return o;
  }
  int64_t integer() {
    ws();
    size_t b = i;
    if (i < s.size() && s[i] == '-') i++;
    while (i < s.size() && s[i] >= '0' && s[i] <= '9') i++;
    LAMP_CHECK(i > b, "tensor list descriptor: expected an integer at byte " << b);
    return std::stoll(s.substr(b, i - b));
  }
  void skip() {   // any value
    char c = peek();
    if (c == '"') { str(); return; }
    if (c == '{') { i++; if (accept('}')) return; do { str(); expect(':'); skip(); } while (accept(',')); expect('}'); return; }
    if (c == '[') { i++; if (accept(']')) return; do { skip(); } while (accept(',')); expect(']'); return; }
    while (i < s.size() && s[i] != ',' && s[i] != '}' && s[i] != ']') i++;   // number / literal
  }
};

TensorDescriptor parse_descriptor(Json& j) {
  TensorDescriptor d;
  j.expect('{');
  if (!j.accept('}')) {
    do {
      const std::string key = j.str();
      j.expect(':');
      if (key == "dims") { j.expect('['); if (!j.accept(']')) { do d.dims.push_back(j.integer()); while (j.accept(',')); j.expect(']'); } }
      else if (key == "dataType") d.dataType = (int)j.integer();
      else if (key == "byteOffset") d.byteOffset = j.integer();
      else if (key == "byteLength") d.byteLength = j.integer();
      else j.skip();
    } while (j.accept(','));
    j.expect('}');
  }
  return d;
}

TensorList parse_tensor_list(const std::string& text) {
  Json j(text);
  TensorList l;
  j.expect('{');
  if (!j.accept('}')) {
    do {
      const std::string key = j.str();
      j.expect(':');
      if (key == "tensors") { j.expect('['); if (!j.accept(']')) { do l.tensors.push_back(parse_descriptor(j)); while (j.accept(',')); j.expect(']'); } }
      else if (key == "location") l.location = j.str();
      else if (key == "byteOffset") l.byteOffset = j.integer();
      else if (key == "byteLength") l.byteLength = j.integer();
      else j.skip();
    } while (j.accept(','));
    j.expect('}');
  }
  // schemas.scala:45-50
  for (auto& t : l.tensors)
    LAMP_CHECK(t.byteOffset + t.byteLength <= l.byteLength, "Some tensor offset+length is out of bound (" << t.byteOffset << ", " << t.byteLength
                                                                                                           << ") total: " << l.byteLength);
  return l;
}

std::string json_string(const std::string& v) {
  std::string o = "\"";
  for (unsigned char c : v) {
    if (c == '"') o += "\\\""; else if (c == '\\') o += "\\\\"; else if (c == '\n') o += "\\n"; else if (c == '\t') o += "\\t";
    else if (c == '\r') o += "\\r"; else if (c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); o += b; } else o += (char)c;
  }
  return o + "\"";
}

// field order = declaration order of the case classes (what jsoniter-scala's derived codec emits), no whitespace
std::string render_tensor_list(const TensorList& l) {
  std::ostringstream o;
  o << "{\"tensors\":[";
  for (size_t t = 0; t < l.tensors.size(); t++) {
    const auto& d = l.tensors[t];
    o << (t ? "," : "") << "{\"dims\":[";
    for (size_t k = 0; k < d.dims.size(); k++) o << (k ? "," : "") << d.dims[k];
    o << "],\"dataType\":" << d.dataType << ",\"byteOffset\":" << d.byteOffset << ",\"byteLength\":" << d.byteLength << "}";
  }
  o << "],\"location\":" << json_string(l.location) << ",\"byteOffset\":" << l.byteOffset << ",\"byteLength\":" << l.byteLength << "}";
  return o.str();
}

std::string base_name(const std::string& p) { const size_t k = p.find_last_of('/'); return k == std::string::npos ? p : p.substr(k + 1); }
std::string dir_name(const std::string& p) { const size_t k = p.find_last_of('/'); return k == std::string::npos ? std::string(".") : (k == 0 ? std::string("/") : p.substr(0, k)); }

std::vector<char> tensor_bytes(const lamp_tensor* t) {
  lamp_tensor* c = nullptr;
  HCALL(lamp_contiguous(&c, t));
  Ten hold(c);
  int64_t n = 0, w = 0;
  HCALL(lamp_tensor_numel(c, &n));
  HCALL(lamp_tensor_element_size(c, &w));
  std::vector<char> buf((size_t)(n * w));
  if (!buf.empty()) HCALL(lamp_copy_to_host(c, buf.data(), buf.size()));
  return buf;
}

void write_tensors(const std::vector<const lamp_tensor*>& ts, const std::string& path) {
  // Writer.writeTensorsIntoFile: the blob goes to "<path>.data", its location is recorded relative to the descriptor
  const std::string data_path = path + ".data";
  std::ofstream blob(data_path, std::ios::binary | std::ios::trunc);
  LAMP_CHECK(blob.good(), "cannot open " << data_path << " for writing");
  TensorList l;
  l.location = base_name(data_path);
  int64_t offset = 0;
  static const char zeros[8] = {0};
  for (const lamp_tensor* t : ts) {
    LAMP_CHECK(t, "writeTensorsIntoFile: NULL tensor");
    TensorDescriptor d;
    int nd = 0, dt = 0;
    int64_t sz[LAMP_MAX_DIMS];
    HCALL(lamp_tensor_ndim(t, &nd));
    HCALL(lamp_tensor_sizes(t, sz));
    HCALL(lamp_tensor_scalar_type(t, &dt));
    d.dims.assign(sz, sz + nd);
    d.dataType = dt;
    const std::vector<char> bytes = tensor_bytes(t);
    d.byteOffset = offset;
    d.byteLength = (int64_t)bytes.size();
    if (!bytes.empty()) blob.write(bytes.data(), (std::streamsize)bytes.size());
    const int64_t pad = (8 - d.byteLength % 8) % 8;      // Writer.scala:112-123: every tensor starts on a multiple of 8
    if (pad) blob.write(zeros, pad);
    offset += d.byteLength + pad;
    l.tensors.push_back(std::move(d));
  }
  l.byteOffset = 0;
  l.byteLength = offset;
  blob.close();
  LAMP_CHECK(!blob.fail(), "write to " << data_path << " failed");
  std::ofstream desc(path, std::ios::binary | std::ios::trunc);
  LAMP_CHECK(desc.good(), "cannot open " << path << " for writing");
  desc << render_tensor_list(l);
  desc.close();
  LAMP_CHECK(!desc.fail(), "write to " << path << " failed");
}

TensorList read_descriptor(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  LAMP_CHECK(f.good(), "cannot open tensor list descriptor " << path);
  std::stringstream ss;
  ss << f.rdbuf();
  return parse_tensor_list(ss.str());
}

Ten empty_tensor(const std::vector<int64_t>& dims, int dtype, int device) {
  lamp_tensor* o = nullptr;
  HCALL(lamp_empty(&o, dims.data(), (int)dims.size(), dtype, device));
  return Ten(o);
}

std::vector<Ten> read_tensors(const std::string& path, int device, bool pin) {
  const TensorList l = read_descriptor(path);
  std::vector<Ten> out;
  if (l.tensors.empty()) return out;
  // STen.tensorsFromFile's contract (STen.scala:157-168)
  LAMP_CHECK(l.byteOffset % 4096 == 0, "Offset must be multiple of 4096. Got " << l.byteOffset << ". Tried to create tensor from " << path << ".");
  for (auto& t : l.tensors) LAMP_CHECK(t.byteOffset % 8 == 0, "Some tensor offsets within the list is not aligned to 8");
  const std::string data_path = (!l.location.empty() && l.location[0] == '/') ? l.location : dir_name(path) + "/" + l.location;
  std::ifstream blob;
  if (l.byteLength > 0) {
    blob.open(data_path, std::ios::binary);
    LAMP_CHECK(blob.good(), "cannot open tensor data " << data_path);
  }
  for (auto& d : l.tensors) {
    int64_t numel = 1;
    for (int64_t v : d.dims) { LAMP_CHECK(v >= 0, "negative dimension in tensor list descriptor"); numel *= v; }
    const size_t width = dtype_size(d.dataType);
    LAMP_CHECK((int64_t)(numel * width) == d.byteLength, "tensor descriptor dims " << numel << " x " << width << " B do not match byteLength " << d.byteLength);
    Ten host = empty_tensor(d.dims, d.dataType, -1);
    if (pin) { lamp_tensor* p = nullptr; HCALL(lamp_pin_memory(&p, host.h())); host = Ten(p); }
    if (d.byteLength > 0) {
      void* dst = nullptr;
      HCALL(lamp_tensor_data_ptr(host.h(), &dst));
      blob.seekg((std::streamoff)(l.byteOffset + d.byteOffset));
      blob.read((char*)dst, (std::streamsize)d.byteLength);
      LAMP_CHECK(blob.gcount() == (std::streamsize)d.byteLength, "tensor data " << data_path << " is shorter than its descriptor says");
    }
    if (device >= 0) {
      lamp_tensor* dv = nullptr;
      HCALL(lamp_to(&dv, host.h(), d.dataType, device, 0, 0));
      out.emplace_back(dv);
    } else out.push_back(host);
  }
  return out;
}

}  // namespace

// ---- the minibatch stream ------------------------------------------------------------------------------------------
struct BatchStreamImpl {
  Ten features, target, order;   // target and order (i64 [n]) on `device`; features on `device`, or - host_resident - in pinned host memory
  int64_t n = 0, minibatch = 1, num_batches = 0, cursor = 0, every = 1, offset = 0;
  // host-resident variant (lamp_batch_stream_from_full_host): the GPU gathers a minibatch's rows over PCIe, queued one batch ahead (on a side stream with LAMP_HOST_STREAM_SIDE=1)
  // (the reference's prefetch: IOLoops.scala:833-874 loads batch i + 1 while batch i trains)
  bool host_resident = false;
  int device = 0, out_dtype = -1;
  lamp_stream* side = nullptr;
  Ten ahead_x, ahead_t;          // the prefetched batch (valid when ahead_for >= 0)
  int64_t ahead_for = -1;
  lamp_stream* ahead_stream = nullptr;   // default mode: the stream the prefetched batch was queued on (the consumer's current stream THEN)
  ~BatchStreamImpl() { if (side) lamp_stream_release(side); if (ahead_stream) lamp_stream_release(ahead_stream); }
};

}  // namespace host
}  // namespace lamp

using namespace lamp;
using namespace lamp::host;

struct lamp_batch_stream { BatchStreamImpl s; };

extern "C" {

int lamp_write_tensors_into_file(lamp_tensor* const* tensors, int64_t n, const char* path) {
  LAMP_API_BEGIN
  LAMP_CHECK(path && (tensors || n == 0), "writeTensorsIntoFile: NULL argument");
  write_tensors(std::vector<const lamp_tensor*>(tensors, tensors + n), path);
  LAMP_API_END
}

int lamp_tensor_list_length(const char* path, int64_t* n) {
  LAMP_API_BEGIN
  *n = (int64_t)read_descriptor(path).tensors.size();
  LAMP_API_END
}

int lamp_read_tensors_from_file(lamp_tensor** out, int64_t capacity, int64_t* n_read, const char* path, int device, int pin) {
  LAMP_API_BEGIN
  std::vector<Ten> ts = read_tensors(path, device, pin != 0);
  LAMP_CHECK((int64_t)ts.size() <= capacity, "readTensorsFromFile: " << ts.size() << " tensors in " << path << " but room for " << capacity);
  for (size_t i = 0; i < ts.size(); i++) HCALL(lamp_tensor_retain(ts[i].h(), &out[i]));
  if (n_read) *n_read = (int64_t)ts.size();
  LAMP_API_END
}

int lamp_module_write_checkpoint(lamp_module* m, const char* path) {
  LAMP_API_BEGIN
  std::vector<Var> st = m->m->state();
  std::vector<const lamp_tensor*> ts;
  for (auto& v : st) ts.push_back(v->value.h());
  write_tensors(ts, path);
  LAMP_API_END
}

int lamp_module_load_from_file(lamp_module* m, const char* path) {
  LAMP_API_BEGIN
  std::vector<Var> st = m->m->state();
  // Load[...] consumes the list in state order, one copyFrom per tensor (Module.scala:103-113, Linear.scala:39-42)
  const int device = st.empty() ? -1 : st[0]->value.device();
  std::vector<Ten> ts = read_tensors(path, device, false);
  LAMP_CHECK(ts.size() >= st.size(), "checkpoint " << path << " holds " << ts.size() << " tensors, the module has " << st.size());
  for (size_t i = 0; i < st.size(); i++) {
    LAMP_CHECK(ts[i].numel() == st[i]->value.numel(), "checkpoint tensor " << i << " has " << ts[i].numel() << " elements, the module's has "
                                                                            << st[i]->value.numel());
    Ten src = ts[i].shape() == st[i]->value.shape() ? ts[i] : ops::view(ts[i], st[i]->value.shape());
    HCALL(lamp_copy_(st[i]->value.h(), src.h(), 0));
  }
  LAMP_API_END
}

int lamp_cifar_load_image_file(lamp_tensor** labels, lamp_tensor** images, const char* path, int64_t num_images, int dtype, int device) {
  LAMP_API_BEGIN
  LAMP_CHECK(num_images >= 0, "numImages must be non-negative");
  constexpr int64_t REC = 3074;   // coarse label, fine label, 3 x 32 x 32 pixel bytes
  std::ifstream f(path, std::ios::binary);
  LAMP_CHECK(f.good(), "cannot open " << path);
  const std::vector<int64_t> shape = {num_images, REC};
  Ten host = empty_tensor(shape, kU8, -1);
  { lamp_tensor* p = nullptr; HCALL(lamp_pin_memory(&p, host.h())); host = Ten(p); }
  void* dst = nullptr;
  HCALL(lamp_tensor_data_ptr(host.h(), &dst));
  f.read((char*)dst, (std::streamsize)(num_images * REC));
  LAMP_CHECK(f.gcount() == (std::streamsize)(num_images * REC), path << " holds fewer than " << num_images << " records of " << REC << " bytes");
  lamp_tensor* dv = nullptr;
  HCALL(lamp_to(&dv, host.h(), kU8, device, 0, 0));
  Ten all(dv);
  Ten lab = ops::cast(ops::select(all, 1, 1), kI64);                                   // all.select(1, 1).castToLong
  Ten img = ops::cast(ops::slice(all, 1, 2, REC, 1), dtype);                           // all.slice(1, 2, 3074, 1) cast to the precision
  Ten img4 = ops::view(img, {num_images, 3, 32, 32});
  HCALL(lamp_tensor_retain(lab.h(), labels));
  HCALL(lamp_tensor_retain(img4.h(), images));
  LAMP_API_END
}

int lamp_batch_stream_from_full(lamp_batch_stream** out, const lamp_tensor* features, const lamp_tensor* target, const int64_t* order, int64_t n,
                                int64_t minibatch_size, int drop_last, int device) {
  LAMP_API_BEGIN
  LAMP_CHECK(features && target && (order || n == 0), "minibatchesFromFull: NULL argument");
  LAMP_CHECK(minibatch_size >= 1, "minibatchSize must be positive");
  LAMP_CHECK(device >= 0, "minibatchesFromFull needs a GPU device");
  int64_t fs[LAMP_MAX_DIMS], ts[LAMP_MAX_DIMS];
  HCALL(lamp_tensor_sizes(features, fs));
  HCALL(lamp_tensor_sizes(target, ts));
  LAMP_CHECK(n <= fs[0] && fs[0] == ts[0], "minibatchesFromFull: features and target disagree on the number of rows, or the order is longer");
  for (int64_t i = 0; i < n; i++) LAMP_CHECK(order[i] >= 0 && order[i] < fs[0], "minibatchesFromFull: order[" << i << "] = " << order[i] << " is out of range");
  auto stream = std::make_unique<lamp_batch_stream>();
  BatchStreamImpl& s = stream->s;
  auto on_device = [&](const lamp_tensor* t) {
    int dt = 0;
    HCALL(lamp_tensor_scalar_type(t, &dt));
    lamp_tensor* o = nullptr;
    HCALL(lamp_to(&o, t, dt, device, 0, 0));
    return Ten(o);
  };
  s.features = on_device(features);
  s.target = on_device(target);
  const std::vector<int64_t> osz = {n};
  s.order = empty_tensor(osz, kI64, device);
  if (n) HCALL(lamp_copy_from_host(s.order.h(), order, (size_t)n * 8));
  s.n = n;
  s.minibatch = minibatch_size;
  // order.grouped(minibatchSize); dropLast removes the last group whether or not it is full (BatchStream.scala:575-585)
  s.num_batches = (n + minibatch_size - 1) / minibatch_size;
  if (drop_last && s.num_batches > 0) s.num_batches--;
  *out = stream.release();
  LAMP_API_END
}

int lamp_batch_stream_from_full_host(lamp_batch_stream** out, const lamp_tensor* features, const lamp_tensor* target, const int64_t* order, int64_t n,
                                     int64_t minibatch_size, int drop_last, int device, int out_dtype) {
  LAMP_API_BEGIN
  LAMP_CHECK(features && target && (order || n == 0), "minibatchesFromFull: NULL argument");
  LAMP_CHECK(minibatch_size >= 1, "minibatchSize must be positive");
  LAMP_CHECK(device >= 0, "minibatchesFromFull needs a GPU device");
  int64_t fs[LAMP_MAX_DIMS], ts[LAMP_MAX_DIMS];
  HCALL(lamp_tensor_sizes(features, fs));
  HCALL(lamp_tensor_sizes(target, ts));
  LAMP_CHECK(n <= fs[0] && fs[0] == ts[0], "minibatchesFromFull: features and target disagree on the number of rows, or the order is longer");
  for (int64_t i = 0; i < n; i++) LAMP_CHECK(order[i] >= 0 && order[i] < fs[0], "minibatchesFromFull: order[" << i << "] = " << order[i] << " is out of range");
  int fdev = 0;
  HCALL(lamp_tensor_device(features, &fdev));
  LAMP_CHECK(fdev < 0, "lamp_batch_stream_from_full_host: the features must live in host memory");
  {
    // the conversions the gather kernel has (kernels/index.hip, lamp_index_select_pinned): checked here, not at the first batch
    int fdt = 0;
    HCALL(lamp_tensor_scalar_type(features, &fdt));
    const int odt = out_dtype < 0 ? fdt : out_dtype;
    const bool ok = odt == fdt || (fdt == kF32 && odt == kBF16) || (fdt == kU8 && (odt == kBF16 || odt == kF32)) || (fdt == kF64 && odt == kF32);
    LAMP_CHECK(ok, "lamp_batch_stream_from_full_host: no gather converts records of scalar type " << fdt << " to " << odt
                   << " (same type, f32 -> bf16, u8 -> bf16 / f32, f64 -> f32)");
  }
  auto stream = std::make_unique<lamp_batch_stream>();
  BatchStreamImpl& s = stream->s;
  int pinned = 0;
  HCALL(lamp_tensor_is_pinned(features, &pinned));
  if (pinned) { lamp_tensor* r = nullptr; HCALL(lamp_tensor_retain(features, &r)); s.features = Ten(r); }
  else { lamp_tensor* p = nullptr; HCALL(lamp_pin_memory(&p, features)); s.features = Ten(p); }      // the reference's `pinned = true` (cifar100.scala --pinned)
  { int dt = 0; HCALL(lamp_tensor_scalar_type(target, &dt)); lamp_tensor* o = nullptr; HCALL(lamp_to(&o, target, dt, device, 0, 0)); s.target = Ten(o); }
  const std::vector<int64_t> osz = {n};
  s.order = empty_tensor(osz, kI64, device);
  if (n) HCALL(lamp_copy_from_host(s.order.h(), order, (size_t)n * 8));
  s.n = n; s.minibatch = minibatch_size; s.device = device; s.out_dtype = out_dtype; s.host_resident = true;
  s.num_batches = (n + minibatch_size - 1) / minibatch_size;
  if (drop_last && s.num_batches > 0) s.num_batches--;
  // LAMP_HOST_STREAM_SIDE=1: the gather on a side stream, ordered by events (round 4's first form).  Measured (bench.py --workload epoch):
  // a kernel that reads host memory does not run beside the training step anyway, so the side stream bought nothing at B = 2048 (1.64 M
  // samples/s either way), cost 12 % at B = 256 (three cross-stream waits per batch) and, for some streams of a process, put the whole
  // epoch into a 3 x slower mode (EXPERIMENTS (17)): the default queues the gather on the consumer's stream.
  static const bool use_side = [] { const char* e = getenv("LAMP_HOST_STREAM_SIDE"); return e && e[0] == '1'; }();
  if (use_side) HCALL(lamp_stream_get_from_pool(0, device, &s.side));
  *out = stream.release();
  LAMP_API_END
}

// queues the gather of batch `b` on the side stream (which first waits for everything the compute stream has queued: the order tensor and,
// through the allocator's stream bookkeeping, the blocks it is about to reuse)
static void prefetch_batch(BatchStreamImpl& s, int64_t b) {
  const int64_t lo = b * s.minibatch, hi = std::min(lo + s.minibatch, s.n);
  Ten idx = ops::slice(s.order, 0, lo, hi, 1);
  lamp_tensor *x = nullptr, *t = nullptr;
  if (!s.side) {                                             // default: the gather is queued on the consumer's stream
    // ... of THIS call: a consumer that has another current stream when the batch is handed out (the reference's loader runs on its own
    // stream, device.scala:199-213) is ordered behind this one then
    if (s.ahead_stream) { (void)lamp_stream_release(s.ahead_stream); s.ahead_stream = nullptr; }
    HCALL(lamp_stream_get_current(s.device, &s.ahead_stream));
    const int rc1 = lamp_index_select_pinned(&x, s.features.h(), idx.h(), s.out_dtype);
    const int rc2 = rc1 == 0 ? lamp_index_select(&t, s.target.h(), 0, idx.h()) : 1;
    if (rc1 != 0 || rc2 != 0) { if (x) lamp_tensor_release(x); throw Error(lamp_last_error()); }
    s.ahead_x = Ten(x); s.ahead_t = Ten(t); s.ahead_for = b;
    return;
  }
  lamp_stream* cur = nullptr;
  HCALL(lamp_stream_get_current(s.device, &cur));
  HCALL(lamp_stream_wait_stream(s.side, cur));
  HCALL(lamp_stream_set_current(s.side));
  const int rc1 = lamp_index_select_pinned(&x, s.features.h(), idx.h(), s.out_dtype);
  const int rc2 = rc1 == 0 ? lamp_index_select(&t, s.target.h(), 0, idx.h()) : 1;
  (void)lamp_stream_set_current(cur);
  (void)lamp_stream_release(cur);
  if (rc1 != 0 || rc2 != 0) { if (x) lamp_tensor_release(x); throw Error(lamp_last_error()); }
  s.ahead_x = Ten(x); s.ahead_t = Ten(t); s.ahead_for = b;
}

int lamp_batch_stream_every_nth(lamp_batch_stream* st, int64_t n, int64_t offset) {
  LAMP_API_BEGIN
  LAMP_CHECK(n >= 1 && offset >= 0 && offset < n, "everyNth(n, offset) needs 0 <= offset < n");
  LAMP_CHECK(st->s.every == 1, "everyNth was already applied to this stream");
  st->s.every = n; st->s.offset = offset;
  LAMP_API_END
}

int lamp_batch_stream_num_batches(const lamp_batch_stream* st, int64_t* out) {
  LAMP_API_BEGIN
  const auto& s = st->s;
  *out = s.num_batches > s.offset ? (s.num_batches - s.offset + s.every - 1) / s.every : 0;
  LAMP_API_END
}

int lamp_batch_stream_next(lamp_batch_stream* st, lamp_tensor** x, lamp_tensor** target) {
  LAMP_API_BEGIN
  auto& s = st->s;
  *x = nullptr; *target = nullptr;
  while (s.cursor < s.num_batches && s.cursor % s.every != s.offset) s.cursor++;
  if (s.cursor >= s.num_batches) return 0;               // EndStream
  if (s.host_resident) {
    const int64_t b = s.cursor++;
    if (s.ahead_for != b) prefetch_batch(s, b);              // first batch of an epoch (or after a reset)
    Ten xb = s.ahead_x, tb = s.ahead_t;
    s.ahead_x = Ten(); s.ahead_t = Ten(); s.ahead_for = -1;
    // the consumer (the caller's current stream) waits for the gather, and the blocks - allocated under the side stream - must not be
    // recycled there while the consumer still reads them
    if (s.side) {
      lamp_stream* cur = nullptr;
      HCALL(lamp_stream_get_current(s.device, &cur));
      HCALL(lamp_stream_wait_stream(cur, s.side));
      HCALL(lamp_tensor_record_stream(xb.h(), cur));
      HCALL(lamp_tensor_record_stream(tb.h(), cur));
      HCALL(lamp_stream_release(cur));
    } else if (s.ahead_stream) {
      // queued on whatever stream was current one call ago: the same one now (the usual case) needs nothing
      lamp_stream* cur = nullptr;
      HCALL(lamp_stream_get_current(s.device, &cur));
      void *a = nullptr, *c = nullptr;
      (void)lamp_stream_native(s.ahead_stream, &a); (void)lamp_stream_native(cur, &c);
      if (a != c) {
        HCALL(lamp_stream_wait_stream(cur, s.ahead_stream));
        HCALL(lamp_tensor_record_stream(xb.h(), cur));
        HCALL(lamp_tensor_record_stream(tb.h(), cur));
      }
      HCALL(lamp_stream_release(cur));
    }
    int64_t nb = s.cursor;                                   // the next batch this stream will hand out
    while (nb < s.num_batches && nb % s.every != s.offset) nb++;
    if (nb < s.num_batches) prefetch_batch(s, nb);
    HCALL(lamp_tensor_retain(xb.h(), x));
    HCALL(lamp_tensor_retain(tb.h(), target));
    return 0;
  }
  const int64_t lo = s.cursor * s.minibatch, hi = std::min(lo + s.minibatch, s.n);
  s.cursor++;
  Ten idx = ops::slice(s.order, 0, lo, hi, 1);
  Ten xb = ops::index_select(s.features, 0, idx), tb = ops::index_select(s.target, 0, idx);
  HCALL(lamp_tensor_retain(xb.h(), x));
  HCALL(lamp_tensor_retain(tb.h(), target));
  LAMP_API_END
}

int lamp_batch_stream_reset(lamp_batch_stream* st) {
  LAMP_API_BEGIN
  st->s.cursor = 0;
  st->s.ahead_x = Ten(); st->s.ahead_t = Ten(); st->s.ahead_for = -1;
  LAMP_API_END
}
int lamp_batch_stream_release(lamp_batch_stream* st) { LAMP_API_BEGIN delete st; LAMP_API_END }

}  // extern "C"
