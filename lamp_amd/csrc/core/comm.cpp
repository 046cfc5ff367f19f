// Collectives over RCCL / xGMI.
//
// Replaces aten.NcclComm.{get_unique_id, comm_init_rank, broadcast, reduce, comm_destroy}
// (reference: lamp-sten/src/main/scala/lamp/STen.scala:629-671, 1902-1908; call sites
// lamp-data/src/main/scala/lamp/data/distributed/package.scala:683-731) and adds all_reduce:
// the reference's root-centric exchange (74 broadcasts + 38 reduces per step) becomes one
// all-reduce of a flat f32 gradient bucket on MI355X's xGMI mesh, with every rank running the
// same optimiser step (see DESIGN.md "data parallel").
//
// One communicator per GPU / process (`torch.distributed`-style launch); the arrays in the
// signatures exist because the reference API lets one thread drive several GPUs with a
// group call - that form is kept (ncclGroupStart/End around the per-communicator calls).
#include "tensor.h"

#include <rccl/rccl.h>

struct lamp_comm {
  ncclComm_t comm = nullptr;
  int nranks = 0, rank = 0, device = 0;
};

namespace {

#define NCCL_CHECK(expr)                                                                          \
  do {                                                                                            \
    ncclResult_t _r = (expr);                                                                     \
    if (_r != ncclSuccess) throw ::lamp::Error(std::string(#expr) + " failed: " + ncclGetErrorString(_r)); \
  } while (0)

ncclDataType_t nccl_type(int dt) {
  switch (dt) {
    case lamp::kF32: return ncclFloat32;
    case lamp::kF64: return ncclFloat64;
    case lamp::kBF16: return ncclBfloat16;
    case lamp::kF16: return ncclFloat16;
    case lamp::kI64: return ncclInt64;
    case lamp::kI32: return ncclInt32;
    case lamp::kU8: case lamp::kBool: return ncclUint8;
  }
  throw lamp::Error(std::string("dtype not supported by RCCL: ") + lamp::dtype_name(dt));
}
ncclRedOp_t nccl_op(int op) {   // ncclRedOp_t numbering, which is what aten.NcclComm.reduce passes through (lamp itself only uses 0)
  switch (op) {
    case 0: return ncclSum;
    case 1: return ncclProd;
    case 2: return ncclMax;
    case 3: return ncclMin;
  }
  throw lamp::Error("reduction op " + std::to_string(op) + " is not one of 0 sum / 1 prod / 2 max / 3 min");
}
// ncclGroupStart / ncclGroupEnd as a scope: an error thrown between the two must not leave the group open (every later RCCL call
// of the thread would queue into it for ever)
struct GroupGuard {
  bool open = false;
  explicit GroupGuard(int n) { if (n > 1) { NCCL_CHECK(ncclGroupStart()); open = true; } }
  void end() { if (open) { open = false; NCCL_CHECK(ncclGroupEnd()); } }
  ~GroupGuard() { if (open) (void)ncclGroupEnd(); }
};
void check_comm_tensor(const lamp_tensor* t, const lamp_comm* c) {
  lamp::check_device_tensor(t, "tensor");
  LAMP_CHECK(c && c->comm, "null communicator");
  LAMP_CHECK(t->is_contiguous(), "collectives need contiguous tensors, got " << t->describe());
  LAMP_CHECK(t->device() == c->device, "tensor is on device " << t->device() << " but the communicator was created on device " << c->device);
}

}  // namespace

using namespace lamp;

extern "C" {

int lamp_comm_get_unique_id(uint8_t* id_out) {
  LAMP_API_BEGIN
  static_assert(sizeof(ncclUniqueId) == LAMP_UNIQUE_ID_BYTES, "unique id size");
  ncclUniqueId id;
  NCCL_CHECK(ncclGetUniqueId(&id));
  memcpy(id_out, &id, sizeof(id));
  LAMP_API_END
}

int lamp_comm_init_rank(lamp_comm** out, int nranks, const uint8_t* id, int rank) {
  LAMP_API_BEGIN
  LAMP_CHECK(nranks > 0 && rank >= 0 && rank < nranks, "bad rank " << rank << " of " << nranks);
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  auto* c = new lamp_comm();
  c->nranks = nranks;
  c->rank = rank;
  c->device = current_device();
  ncclResult_t r = ncclCommInitRank(&c->comm, nranks, uid, rank);  // blocks until the clique is complete
  if (r != ncclSuccess) {
    delete c;
    throw Error(std::string("ncclCommInitRank failed: ") + ncclGetErrorString(r));
  }
  *out = c;
  LAMP_API_END
}

int lamp_comm_broadcast(lamp_tensor* const* tensors, lamp_comm* const* comms, int n, int root) {
  LAMP_API_BEGIN
  for (int i = 0; i < n; i++) check_comm_tensor(tensors[i], comms[i]);
  GroupGuard group(n);
  for (int i = 0; i < n; i++) {
    NCCL_CHECK(ncclBroadcast(tensors[i]->data(), tensors[i]->data(), (size_t)tensors[i]->numel(), nccl_type(tensors[i]->dtype), root,
                             comms[i]->comm, current_stream(comms[i]->device)));
  }
  group.end();
  LAMP_API_END
}

int lamp_comm_reduce(lamp_tensor* const* inputs, lamp_tensor* output, int root, int op, lamp_comm* const* comms, int n) {
  LAMP_API_BEGIN
  const ncclRedOp_t rop = nccl_op(op);
  for (int i = 0; i < n; i++) {
    check_comm_tensor(inputs[i], comms[i]);
    if (comms[i]->rank == root && output)
      LAMP_CHECK(output->numel() == inputs[i]->numel() && output->dtype == inputs[i]->dtype && output->is_contiguous(), "reduce: output does not match the input");
  }
  GroupGuard group(n);
  for (int i = 0; i < n; i++) {
    void* recv = (comms[i]->rank == root && output) ? output->data() : inputs[i]->data();
    NCCL_CHECK(ncclReduce(inputs[i]->data(), recv, (size_t)inputs[i]->numel(), nccl_type(inputs[i]->dtype), rop, root,
                          comms[i]->comm, current_stream(comms[i]->device)));
  }
  group.end();
  LAMP_API_END
}

int lamp_comm_all_reduce(lamp_tensor* const* tensors, lamp_comm* const* comms, int n, int op) {
  LAMP_API_BEGIN
  const ncclRedOp_t rop = nccl_op(op);
  for (int i = 0; i < n; i++) check_comm_tensor(tensors[i], comms[i]);
  // bench.py's untimed pass brackets the collective like any kernel class ("allreduce_us"); one bracket, on the first stream
  double bytes = 0;
  for (int i = 0; i < n; i++) bytes += (double)tensors[i]->numel() * (double)tensors[i]->itemsize();
  KernelTimer kt("rccl_all_reduce", 0, bytes, n > 0 ? current_stream(comms[0]->device) : nullptr);
  GroupGuard group(n);
  for (int i = 0; i < n; i++) {
    NCCL_CHECK(ncclAllReduce(tensors[i]->data(), tensors[i]->data(), (size_t)tensors[i]->numel(), nccl_type(tensors[i]->dtype), rop,
                             comms[i]->comm, current_stream(comms[i]->device)));
  }
  group.end();
  LAMP_API_END
}

// out[rank * numel(in) ...] = in of that rank (ncclAllGather): the exchange step of row-sharded work (kNN graph)
int lamp_comm_all_gather(lamp_tensor* out, const lamp_tensor* in, lamp_comm* comm) {
  LAMP_API_BEGIN
  check_comm_tensor(out, comm);
  check_comm_tensor(in, comm);
  int nranks = 0;
  NCCL_CHECK(ncclCommCount(comm->comm, &nranks));
  LAMP_CHECK(out->dtype == in->dtype && out->numel() == in->numel() * nranks, "all_gather: out must hold nranks x in elements of the same dtype");
  NCCL_CHECK(ncclAllGather(in->data(), out->data(), (size_t)in->numel(), nccl_type(in->dtype), comm->comm, current_stream(comm->device)));
  LAMP_API_END
}

// ranks RCCL itself counts in the communicator / this rank's index in it (ncclCommCount, ncclCommUserRank): launchers use it to
// prove that an "N GPU" job really is N ranks
int lamp_comm_count(const lamp_comm* c, int* nranks_out) {
  LAMP_API_BEGIN
  LAMP_CHECK(c && c->comm, "null communicator");
  NCCL_CHECK(ncclCommCount(c->comm, nranks_out));
  LAMP_API_END
}
int lamp_comm_user_rank(const lamp_comm* c, int* rank_out) {
  LAMP_API_BEGIN
  LAMP_CHECK(c && c->comm, "null communicator");
  NCCL_CHECK(ncclCommUserRank(c->comm, rank_out));
  LAMP_API_END
}

int lamp_comm_destroy(lamp_comm* c) {
  LAMP_API_BEGIN
  if (c) {
    if (c->comm) (void)ncclCommDestroy(c->comm);
    delete c;
  }
  LAMP_API_END
}

}  // extern "C"
