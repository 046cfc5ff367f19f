#!/usr/bin/env python3
"""JNI adapter generator: include/lamp_hip.h  ->  jni/aten_jni.c + jni/LampNative.java, and the aten name map check.

lamp binds its tensor backend through the JVM package `aten` (aten-scala-core, build.sbt:125).  The drop-in for that seam is
liblamp_hip.so behind a JNI shim.  This script writes the shim mechanically, one native per C-ABI function:

  python scripts/gen_jni.py emit      regenerate jni/aten_jni.c and jni/LampNative.java from include/lamp_hip.h
  python scripts/gen_jni.py collect   (build container only: needs /root/reference) re-grep the names lamp's hot-path modules call on
                                      aten.{ATen, Tensor, TensorOptions, CudaStream, NcclComm, TensorTrace} -> tests/golden/aten_surface.json
  python scripts/gen_jni.py check     every collected name is mapped to an exported symbol or listed as an explicit gap (jni/name_map.json)

Conventions of the generated natives (class aten.LampNative, all static):
  * handles (lamp_tensor*, lamp_stream*, lamp_comm*, lamp_graph*) cross as `long`; 0 = NULL (optional arguments);
  * an `out` handle becomes the return value; functions with several outputs (`out3[3]`, two or more `**`) return long[];
  * (pointer, count) pairs - `const int64_t* sizes, int ndim`, `lamp_tensor* const* ts, int n` - are ONE Java array;
  * `int* / int64_t* / double*` result parameters are returned (long / double), `uint8_t mask[3]` is boolean[];
  * a non-zero status raises java.lang.RuntimeException(lamp_last_error()) - lamp's Scope relies on exceptions propagating
    (Scope.scala:394-421);
  * host buffers (lamp_copy_from_host / lamp_copy_to_host) get one native per primitive array type: the shapes of
    Tensor.copyFrom{Double,Float,Long,Int,Short,Byte}Array / copyTo...Array (TensorHelpers.scala:57, 223, 253).
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "lamp_hip.h")
OUT_C = os.path.join(ROOT, "jni", "aten_jni.c")
OUT_JAVA = os.path.join(ROOT, "jni", "LampNative.java")
SURFACE = os.path.join(ROOT, "tests", "golden", "aten_surface.json")
NAME_MAP = os.path.join(ROOT, "jni", "name_map.json")
HANDLES = ("lamp_tensor", "lamp_stream", "lamp_comm", "lamp_graph")
MANUAL = {"lamp_copy_from_host", "lamp_copy_to_host", "lamp_from_blob", "lamp_tensor_data_ptr", "lamp_stream_native", "lamp_last_error",
          "lamp_version", "lamp_kernel_timer_report", "lamp_device_name", "lamp_comm_get_unique_id", "lamp_comm_init_rank", "lamp_tensor_release_all",
          "lamp_debug_ig8d_stamps", "lamp_tensor_sizes", "lamp_tensor_strides", "lamp_tensors_from_file", "lamp_tensor_trace_list"}


def parse_header():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    src = re.sub(r"#[^\n]*", " ", src)
    out = []
    for m in re.finditer(r"\b(int|const char\*)\s+(lamp_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        params = " ".join(m.group(3).split())
        plist = [] if params in ("", "void") else [p.strip() for p in params.split(",")]
        out.append((m.group(2), m.group(1), plist))
    return out


def split_param(p):
    """-> (type string without the name, name)"""
    arr = ""
    if "[" in p:
        arr = p[p.index("["):]
        p = p[:p.index("[")].strip()
    m = re.match(r"^(.*?)(\w+)$", p)
    t, name = m.group(1).strip(), m.group(2)
    return (t + arr).replace(" *", "*"), name


class P:   # one Java-side parameter or result derived from one or two C parameters
    def __init__(self, kind, cname, jtype=None, jni=None, count_name=None, n=None):
        self.kind, self.cname, self.jtype, self.jni, self.count_name, self.n = kind, cname, jtype, jni, count_name, n


def plan(name, plist):
    """classify the C parameters: inputs (Java parameters) and outputs (Java result)"""
    ps = [split_param(p) for p in plist]
    ins, outs, i = [], [], 0
    while i < len(ps):
        t, n = ps[i]
        nxt = ps[i + 1] if i + 1 < len(ps) else None
        is_count = nxt is not None and nxt[0] == "int" and re.match(r"^(n|ndim|ndims|nd|nreplicas|ntensors|nvars|count|k)\w*$", nxt[1])
        hm = re.match(r"^(const )?(lamp_\w+)\*\*$", t)
        ha = re.match(r"^(const )?(lamp_\w+)\* ?const\*$", t)
        h1 = re.match(r"^(const )?(lamp_\w+)\*$", t)
        oa = re.match(r"^(lamp_\w+)\*\[(\d+)\]$", t)
        if hm and hm.group(2) in HANDLES and re.match(r"^(outs\w*|parts|pieces)$", n):
            # a caller-sized array of result handles (lamp_chunk_contiguous(outs, x, n, dim)): its length is the function's count parameter.
            # classified as ONE out parameter this wrote n handles through the address of a single stack slot
            cnt = [q[1] for q in ps if q[0] == "int" and re.match(r"^(n|count|ntensors|nparts|chunks)$", q[1])]
            if len(cnt) != 1:
                return None
            outs.append(P("handle_outs", n, count_name=cnt[0]))
        elif hm and hm.group(2) in HANDLES:
            outs.append(P("handle", n))
        elif oa:
            outs.append(P("handle_array", n, n=int(oa.group(2))))
        elif ha and ha.group(2) in HANDLES:
            if is_count:
                ins.append(P("handles+count", n, "long[]", "jlongArray", count_name=nxt[1])); i += 1
            else:
                ins.append(P("handles", n, "long[]", "jlongArray"))
        elif h1 and h1.group(2) in HANDLES:
            ins.append(P("handle", n, "long", "jlong"))
        elif t == "const int64_t*":
            if is_count:
                ins.append(P("longs+count", n, "long[]", "jlongArray", count_name=nxt[1])); i += 1
            else:
                ins.append(P("longs", n, "long[]", "jlongArray"))
        elif t == "const double*":
            ins.append(P("doubles", n, "double[]", "jdoubleArray"))
        elif re.match(r"^const uint8_t\[\d+\]$", t):
            ins.append(P("mask", n, "boolean[]", "jbooleanArray", n=int(re.search(r"\[(\d+)\]", t).group(1))))
        elif t in ("int*", "int64_t*", "double*", "uint64_t*"):
            outs.append(P("scalar:" + t[:-1], n))
        elif t == "int":
            ins.append(P("int", n, "int", "jint"))
        elif t in ("int64_t", "uint64_t", "size_t"):
            ins.append(P("long", n, "long", "jlong"))
        elif t == "double":
            ins.append(P("double", n, "double", "jdouble"))
        elif t == "const char*":
            ins.append(P("string", n, "String", "jstring"))
        else:
            return None   # needs a hand-written native
        i += 1
    return ins, outs


def result_types(outs):
    if not outs:
        return "void", "void"
    if len(outs) == 1:
        o = outs[0]
        if o.kind == "handle":
            return "long", "jlong"
        if o.kind in ("handle_array", "handle_outs"):
            return "long[]", "jlongArray"
        return {"scalar:int": ("int", "jint"), "scalar:int64_t": ("long", "jlong"), "scalar:uint64_t": ("long", "jlong"),
                "scalar:double": ("double", "jdouble")}[o.kind]
    if any(o.kind == "handle_outs" for o in outs):
        return None, None                      # a variable-length result next to other results: needs a hand-written native
    if all(o.kind in ("handle", "handle_array") for o in outs):
        return "long[]", "jlongArray"
    if all(o.kind.startswith("scalar:int") or o.kind.startswith("scalar:uint") for o in outs):
        return "long[]", "jlongArray"
    return None, None


def emit_function(name, plist):
    pl = plan(name, plist)
    if pl is None:
        return None, None
    ins, outs = pl
    jret, cret = result_types(outs)
    if jret is None:
        return None, None
    java = f"  public static native {jret} {name}({', '.join(f'{p.jtype} {p.cname}' for p in ins)});"
    c = [f"JNIEXPORT {cret} JNICALL Java_aten_LampNative_{name.replace('_', '_1')}(JNIEnv* env, jclass cls" +
         "".join(f", {p.jni} {p.cname}" for p in ins) + ") {", "  (void)cls;"]
    pre, post, args_by_name = [], [], {}
    for p in ins:
        k = p.kind
        if k == "handle":
            args_by_name[p.cname] = f"H({p.cname})"
        elif k in ("handles", "handles+count"):
            pre.append(f"  jsize {p.cname}_n = (*env)->GetArrayLength(env, {p.cname});")
            pre.append(f"  jlong* {p.cname}_e = (*env)->GetLongArrayElements(env, {p.cname}, NULL);")
            post.append(f"  (*env)->ReleaseLongArrayElements(env, {p.cname}, {p.cname}_e, JNI_ABORT);")
            args_by_name[p.cname] = f"(void*){p.cname}_e"      # jlong and a pointer are both 8 bytes on the LP64 targets of this library
            if p.count_name:
                args_by_name[p.count_name] = f"(int){p.cname}_n"
            else:
                pre.append(f"  (void){p.cname}_n;")
        elif k in ("longs", "longs+count"):
            pre.append(f"  jsize {p.cname}_n = (*env)->GetArrayLength(env, {p.cname});")
            pre.append(f"  jlong* {p.cname}_e = (*env)->GetLongArrayElements(env, {p.cname}, NULL);")
            post.append(f"  (*env)->ReleaseLongArrayElements(env, {p.cname}, {p.cname}_e, JNI_ABORT);")
            args_by_name[p.cname] = f"(const int64_t*){p.cname}_e"
            if p.count_name:
                args_by_name[p.count_name] = f"(int){p.cname}_n"
            else:
                pre.append(f"  (void){p.cname}_n;")
        elif k == "doubles":
            pre.append(f"  jdouble* {p.cname}_e = (*env)->GetDoubleArrayElements(env, {p.cname}, NULL);")
            post.append(f"  (*env)->ReleaseDoubleArrayElements(env, {p.cname}, {p.cname}_e, JNI_ABORT);")
            args_by_name[p.cname] = f"(const double*){p.cname}_e"
        elif k == "mask":
            pre.append(f"  jboolean* {p.cname}_e = (*env)->GetBooleanArrayElements(env, {p.cname}, NULL);")
            pre.append(f"  uint8_t {p.cname}_m[{p.n}]; for (int i_ = 0; i_ < {p.n}; i_++) {p.cname}_m[i_] = {p.cname}_e[i_] ? 1 : 0;")
            post.append(f"  (*env)->ReleaseBooleanArrayElements(env, {p.cname}, {p.cname}_e, JNI_ABORT);")
            args_by_name[p.cname] = f"{p.cname}_m"
        elif k == "string":
            pre.append(f"  const char* {p.cname}_s = {p.cname} ? (*env)->GetStringUTFChars(env, {p.cname}, NULL) : NULL;")
            post.append(f"  if ({p.cname}_s) (*env)->ReleaseStringUTFChars(env, {p.cname}, {p.cname}_s);")
            args_by_name[p.cname] = f"{p.cname}_s"
        elif k == "int":
            args_by_name[p.cname] = f"(int){p.cname}"
        elif k == "long":
            args_by_name[p.cname] = f"(int64_t){p.cname}"
        else:
            args_by_name[p.cname] = f"(double){p.cname}"
    for o in outs:
        if o.kind == "handle":
            pre.append(f"  void* {o.cname}_o = NULL;")
            args_by_name[o.cname] = f"(void*)&{o.cname}_o"
        elif o.kind == "handle_array":
            pre.append(f"  void* {o.cname}_o[{o.n}] = {{0}};")
            args_by_name[o.cname] = f"(void*){o.cname}_o"
        elif o.kind == "handle_outs":
            pre.append(f"  if ({o.count_name} < 0 || {o.count_name} > LAMP_JNI_MAX_OUTS) {{ lamp_throw_msg(env, \"{name}: result count out of range\"); return NULL; }}")
            pre.append(f"  void* {o.cname}_o[LAMP_JNI_MAX_OUTS];")
            pre.append(f"  for (int i_ = 0; i_ < (int){o.count_name}; i_++) {o.cname}_o[i_] = NULL;")
            args_by_name[o.cname] = f"(void*){o.cname}_o"
        else:
            ct = o.kind.split(":")[1]
            pre.append(f"  {ct} {o.cname}_o = 0;")
            args_by_name[o.cname] = f"&{o.cname}_o"
    # C argument list in declaration order
    cargs = []
    for p in plist:
        _, n = split_param(p)
        cargs.append(args_by_name[n])
    c += pre
    c.append(f"  const int rc_ = {name}({', '.join(cargs)});")
    c += post
    fail = {"void": "return;", "jlong": "return 0;", "jint": "return 0;", "jdouble": "return 0;", "jlongArray": "return NULL;"}[cret]
    c.append(f"  if (rc_ != 0) {{ lamp_throw(env); {fail} }}")
    if cret == "void":
        pass
    elif len(outs) == 1 and outs[0].kind == "handle":
        c.append(f"  return (jlong)(intptr_t){outs[0].cname}_o;")
    elif len(outs) == 1 and outs[0].kind.startswith("scalar"):
        c.append(f"  return ({cret}){outs[0].cname}_o;")
    elif len(outs) == 1 and outs[0].kind == "handle_outs":
        o = outs[0]
        c.append(f"  jlongArray a_ = (*env)->NewLongArray(env, (jsize){o.count_name});")
        c.append(f"  if (a_) {{ jlong r_[LAMP_JNI_MAX_OUTS]; for (int i_ = 0; i_ < (int){o.count_name}; i_++) r_[i_] = (jlong)(intptr_t){o.cname}_o[i_]; "
                 f"(*env)->SetLongArrayRegion(env, a_, 0, (jsize){o.count_name}, r_); }}")
        c.append("  return a_;")
    else:
        vals = []
        for o in outs:
            if o.kind == "handle":
                vals.append(f"(jlong)(intptr_t){o.cname}_o")
            elif o.kind == "handle_array":
                vals += [f"(jlong)(intptr_t){o.cname}_o[{k}]" for k in range(o.n)]
            else:
                vals.append(f"(jlong){o.cname}_o")
        c.append(f"  jlong r_[{len(vals)}] = {{{', '.join(vals)}}};")
        c.append(f"  jlongArray a_ = (*env)->NewLongArray(env, {len(vals)});")
        c.append(f"  if (a_) (*env)->SetLongArrayRegion(env, a_, 0, {len(vals)}, r_);")
        c.append("  return a_;")
    c.append("}")
    return java, "\n".join(c)


C_PROLOGUE = '''/* GENERATED by scripts/gen_jni.py from include/lamp_hip.h - do not edit.
 *
 * JNI shim of the class aten.LampNative over liblamp_hip.so: the adapter a JVM deployment of lamp loads instead of aten-scala's
 * libatenscalajni (build.sbt:125).  Build where a JDK exists:
 *   cc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude jni/aten_jni.c -Llamp_amd/lib -llamp_hip -o liblampjni.so
 * In images without a JDK, `cc -fsyntax-only -DLAMP_JNI_SYNTAX_CHECK -Iinclude -Ijni jni/aten_jni.c` checks it against jni/jni_syntax_check.h
 * (declarations only, NOT ABI compatible with a real jni.h).
 */
#ifdef LAMP_JNI_SYNTAX_CHECK
#include "jni_syntax_check.h"
#else
#include <jni.h>
#endif
#include <stdint.h>
#include <stddef.h>
#include "lamp_hip.h"

#define H(x) ((void*)(intptr_t)(x))

#define LAMP_JNI_MAX_OUTS 4096   /* bound of a caller-sized array of result handles (lamp_chunk_contiguous, lamp_tensors_from_file) */

static void lamp_throw(JNIEnv* env) {
  jclass ex = (*env)->FindClass(env, "java/lang/RuntimeException");
  if (ex) (*env)->ThrowNew(env, ex, lamp_last_error());
}
static void lamp_throw_msg(JNIEnv* env, const char* msg) {
  jclass ex = (*env)->FindClass(env, "java/lang/RuntimeException");
  if (ex) (*env)->ThrowNew(env, ex, msg);
}

/* ---- hand-written natives: host buffers, strings, raw pointers ------------------------------------------------------------------ */
#define COPY_NATIVES(JT, JNAME, GET, REL)                                                                                              \\
  JNIEXPORT jboolean JNICALL Java_aten_LampNative_copyFrom##JNAME##Array(JNIEnv* env, jclass cls, jlong t, JT##Array a) {              \\
    (void)cls;                                                                                                                         \\
    jsize n = (*env)->GetArrayLength(env, a);                                                                                         \\
    JT* e = (*env)->GET(env, a, NULL);                                                                                                \\
    const int rc_ = lamp_copy_from_host(H(t), e, (size_t)n * sizeof(JT));                                                             \\
    (*env)->REL(env, a, e, JNI_ABORT);                                                                                                \\
    return rc_ == 0 ? JNI_TRUE : JNI_FALSE;                                                                                           \\
  }                                                                                                                                    \\
  JNIEXPORT jboolean JNICALL Java_aten_LampNative_copyTo##JNAME##Array(JNIEnv* env, jclass cls, jlong t, JT##Array a) {                \\
    (void)cls;                                                                                                                         \\
    jsize n = (*env)->GetArrayLength(env, a);                                                                                         \\
    JT* e = (*env)->GET(env, a, NULL);                                                                                                \\
    const int rc_ = lamp_copy_to_host(H(t), e, (size_t)n * sizeof(JT));                                                               \\
    (*env)->REL(env, a, e, rc_ == 0 ? 0 : JNI_ABORT);                                                                                 \\
    return rc_ == 0 ? JNI_TRUE : JNI_FALSE;                                                                                           \\
  }
COPY_NATIVES(jdouble, Double, GetDoubleArrayElements, ReleaseDoubleArrayElements)
COPY_NATIVES(jfloat, Float, GetFloatArrayElements, ReleaseFloatArrayElements)
COPY_NATIVES(jlong, Long, GetLongArrayElements, ReleaseLongArrayElements)
COPY_NATIVES(jint, Int, GetIntArrayElements, ReleaseIntArrayElements)
COPY_NATIVES(jshort, Short, GetShortArrayElements, ReleaseShortArrayElements)
COPY_NATIVES(jbyte, Byte, GetByteArrayElements, ReleaseByteArrayElements)

JNIEXPORT jstring JNICALL Java_aten_LampNative_lamp_1last_1error(JNIEnv* env, jclass cls) { (void)cls; return (*env)->NewStringUTF(env, lamp_last_error()); }
JNIEXPORT jstring JNICALL Java_aten_LampNative_lamp_1version(JNIEnv* env, jclass cls) { (void)cls; return (*env)->NewStringUTF(env, lamp_version()); }
JNIEXPORT void JNICALL Java_aten_LampNative_lamp_1tensor_1release_1all(JNIEnv* env, jclass cls, jlongArray ts) {   /* Tensor.releaseAll */
  (void)cls;
  jsize n = (*env)->GetArrayLength(env, ts);
  jlong* e = (*env)->GetLongArrayElements(env, ts, NULL);
  const int rc_ = lamp_tensor_release_all((void*)e, (int)n);
  (*env)->ReleaseLongArrayElements(env, ts, e, JNI_ABORT);
  if (rc_ != 0) lamp_throw(env);
}
JNIEXPORT jbyteArray JNICALL Java_aten_LampNative_lamp_1comm_1get_1unique_1id(JNIEnv* env, jclass cls) {   /* NcclComm.get_unique_id */
  (void)cls;
  uint8_t id[LAMP_UNIQUE_ID_BYTES];
  if (lamp_comm_get_unique_id(id) != 0) { lamp_throw(env); return NULL; }
  jbyteArray a = (*env)->NewByteArray(env, LAMP_UNIQUE_ID_BYTES);
  if (a) (*env)->SetByteArrayRegion(env, a, 0, LAMP_UNIQUE_ID_BYTES, (const jbyte*)id);
  return a;
}
JNIEXPORT jlong JNICALL Java_aten_LampNative_lamp_1comm_1init_1rank(JNIEnv* env, jclass cls, jint nranks, jbyteArray id, jint rank) {   /* NcclComm.comm_init_rank */
  (void)cls;
  jbyte* e = (*env)->GetByteArrayElements(env, id, NULL);
  lamp_comm* c = NULL;
  const int rc_ = lamp_comm_init_rank(&c, (int)nranks, (const uint8_t*)e, (int)rank);
  (*env)->ReleaseByteArrayElements(env, id, e, JNI_ABORT);
  if (rc_ != 0) { lamp_throw(env); return 0; }
  return (jlong)(intptr_t)c;
}

static jlongArray lamp_dims(JNIEnv* env, jlong t, int strides) {   /* Tensor.sizes() / strides(): long[ndim] */
  int nd = 0;
  int64_t v[LAMP_MAX_DIMS];
  if (lamp_tensor_ndim(H(t), &nd) != 0 || (strides ? lamp_tensor_strides(H(t), v) : lamp_tensor_sizes(H(t), v)) != 0) { lamp_throw(env); return NULL; }
  jlongArray a = (*env)->NewLongArray(env, nd);
  if (a) (*env)->SetLongArrayRegion(env, a, 0, nd, (const jlong*)v);
  return a;
}
JNIEXPORT jlongArray JNICALL Java_aten_LampNative_lamp_1tensor_1sizes(JNIEnv* env, jclass cls, jlong t) { (void)cls; return lamp_dims(env, t, 0); }
JNIEXPORT jlongArray JNICALL Java_aten_LampNative_lamp_1tensor_1strides(JNIEnv* env, jclass cls, jlong t) { (void)cls; return lamp_dims(env, t, 1); }

JNIEXPORT jlongArray JNICALL Java_aten_LampNative_lamp_1tensors_1from_1file(JNIEnv* env, jclass cls, jstring path, jlong offset, jlong length, jboolean pin,
                                                                            jlongArray types, jlongArray offsets, jlongArray lengths) {   /* Tensor.tensors_from_file */
  (void)cls;
  const jsize n = (*env)->GetArrayLength(env, types);
  if (n > 4096) { jclass ex = (*env)->FindClass(env, "java/lang/IllegalArgumentException"); if (ex) (*env)->ThrowNew(env, ex, "more than 4096 tensors in one list"); return NULL; }
  const char* p = (*env)->GetStringUTFChars(env, path, NULL);
  jlong* t = (*env)->GetLongArrayElements(env, types, NULL);
  jlong* o = (*env)->GetLongArrayElements(env, offsets, NULL);
  jlong* l = (*env)->GetLongArrayElements(env, lengths, NULL);
  lamp_tensor* outs[4096];
  const int rc_ = lamp_tensors_from_file(outs, p, (int64_t)offset, (int64_t)length, pin ? 1 : 0, (const int64_t*)t, (const int64_t*)o, (const int64_t*)l, (int)n);
  (*env)->ReleaseLongArrayElements(env, lengths, l, JNI_ABORT);
  (*env)->ReleaseLongArrayElements(env, offsets, o, JNI_ABORT);
  (*env)->ReleaseLongArrayElements(env, types, t, JNI_ABORT);
  (*env)->ReleaseStringUTFChars(env, path, p);
  if (rc_ != 0) { lamp_throw(env); return NULL; }
  jlongArray a = (*env)->NewLongArray(env, n);
  if (a) { jlong h[4096]; for (jsize i = 0; i < n; i++) h[i] = (jlong)(intptr_t)outs[i]; (*env)->SetLongArrayRegion(env, a, 0, n, h); }
  return a;
}
JNIEXPORT jlongArray JNICALL Java_aten_LampNative_lamp_1tensor_1trace_1list(JNIEnv* env, jclass cls) {   /* TensorTrace.list: LAMP_TRACE_RECORD longs per live handle */
  (void)cls;
  int64_t count = 0;
  if (lamp_tensor_trace_list(NULL, 0, &count) != 0) { lamp_throw(env); return NULL; }
  jlongArray a = (*env)->NewLongArray(env, (jsize)(count * LAMP_TRACE_RECORD));
  if (!a || count == 0) return a;
  jlong* e = (*env)->GetLongArrayElements(env, a, NULL);
  int64_t got = 0;
  const int rc_ = lamp_tensor_trace_list((int64_t*)e, count, &got);
  (*env)->ReleaseLongArrayElements(env, a, e, 0);
  if (rc_ != 0) { lamp_throw(env); return NULL; }
  return a;
}

/* ---- generated natives ------------------------------------------------------------------------------------------------------------ */
'''

JAVA_PROLOGUE = '''// GENERATED by scripts/gen_jni.py from include/lamp_hip.h - do not edit.
package aten;

/** Natives of liblamp_hip.so (through liblampjni.so).  Handles are longs; 0 is NULL.  See jni/name_map.json for the aten.* name each
 *  function stands behind. */
public final class LampNative {
  static { System.loadLibrary("lampjni"); }
  private LampNative() {}
  public static native String lamp_last_error();
  public static native String lamp_version();
  public static native void lamp_tensor_release_all(long[] tensors);
  public static native byte[] lamp_comm_get_unique_id();
  public static native long lamp_comm_init_rank(int nranks, byte[] id, int rank);
  public static native long[] lamp_tensors_from_file(String path, long offset, long length, boolean pin, long[] types, long[] offsets, long[] lengths);
  public static native long[] lamp_tensor_trace_list();
  public static native long[] lamp_tensor_sizes(long t);
  public static native long[] lamp_tensor_strides(long t);
  public static native boolean copyFromDoubleArray(long t, double[] a);
  public static native boolean copyToDoubleArray(long t, double[] a);
  public static native boolean copyFromFloatArray(long t, float[] a);
  public static native boolean copyToFloatArray(long t, float[] a);
  public static native boolean copyFromLongArray(long t, long[] a);
  public static native boolean copyToLongArray(long t, long[] a);
  public static native boolean copyFromIntArray(long t, int[] a);
  public static native boolean copyToIntArray(long t, int[] a);
  public static native boolean copyFromShortArray(long t, short[] a);
  public static native boolean copyToShortArray(long t, short[] a);
  public static native boolean copyFromByteArray(long t, byte[] a);
  public static native boolean copyToByteArray(long t, byte[] a);
'''


def generate():
    java, c, skipped = [JAVA_PROLOGUE], [C_PROLOGUE], []
    for name, ret, plist in parse_header():
        if name in MANUAL:
            continue
        j, cc = (None, None) if ret != "int" else emit_function(name, plist)
        if j is None:
            skipped.append(name)
            continue
        java.append(j)
        c.append(cc + "\n")
    java.append("}\n")
    c.append("/* functions of lamp_hip.h without a generated native (raw host pointers / debugging): " + ", ".join(sorted(skipped)) + " */\n")
    return "\n".join(java), "\n".join(c), skipped


def exported_symbols():
    import subprocess
    lib = os.path.join(ROOT, "lamp_amd", "lib", "liblamp_hip.so")
    out = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    return {l.split()[-1] for l in out.splitlines() if l.split()[-1].startswith("lamp_")}


REF_DIRS = ["lamp-sten/src/main", "lamp-core/src/main/scala/lamp/autograd", "lamp-core/src/main/scala/lamp/nn", "lamp-data/src/main",
            "lamp-knn/src/main", "lamp-umap/src/main"]


def collect():
    import glob
    ref = "/root/reference"
    text = ""
    for d in REF_DIRS:
        for f in glob.glob(os.path.join(ref, d, "**", "*.scala"), recursive=True):
            text += open(f).read() + "\n"
    surf = {}
    for cls in ("ATen", "Tensor", "TensorOptions", "CudaStream", "NcclComm", "TensorTrace", "TensorOptionsTrace", "NcclUniqueId"):
        surf[cls] = sorted(set(re.findall(r"\b" + cls + r"\s*\.\s*([A-Za-z_][A-Za-z_0-9]*)", text)) - {"apply", "type", "scala"})
    # instance methods called on the `value` of an STen / on aten.Tensor values: `.value.xyz(` inside lamp-sten
    sten = "".join(open(f).read() for f in glob.glob(os.path.join(ref, "lamp-sten/src/main", "**", "*.scala"), recursive=True))
    surf["Tensor instance"] = sorted(set(re.findall(r"\bvalue\.([a-zA-Z_][A-Za-z_0-9]*)\s*[\(\n ]", sten)) - {"value"})
    surf["_source"] = {"dirs": REF_DIRS, "note": "names only, collected with scripts/gen_jni.py collect in the build container"}
    os.makedirs(os.path.dirname(SURFACE), exist_ok=True)
    json.dump(surf, open(SURFACE, "w"), indent=1)
    print({k: len(v) for k, v in surf.items() if k != "_source"})


def base_name(n):
    """ATen.add_0_l -> add ; fill__0 -> fill_ ; norm_2 -> norm ; _cast_Double stays"""
    b = re.sub(r"_l$", "", n)
    b = re.sub(r"_(\d+)$", "", b)
    return b


def check():
    surf = json.load(open(SURFACE))
    nm = json.load(open(NAME_MAP))
    syms = exported_symbols()
    problems, mapped, gaps = [], 0, 0
    for cls, names in surf.items():
        if cls.startswith("_"):
            continue
        table = nm.get(cls, {})
        for n in names:
            e = table.get(n) or table.get(base_name(n))
            if e is None:
                guess = "lamp_" + base_name(n).lstrip("_")
                if cls == "ATen" and guess in syms:
                    mapped += 1
                    continue
                problems.append(f"{cls}.{n}: neither mapped nor listed as a gap")
            elif "symbol" in e:
                for s in ([e["symbol"]] if isinstance(e["symbol"], str) else e["symbol"]):
                    if s not in syms and not s.startswith("jvm:"):
                        problems.append(f"{cls}.{n} -> {s}: not exported by liblamp_hip.so")
                mapped += 1
            else:
                assert "gap" in e, (cls, n)
                gaps += 1
    return mapped, gaps, problems


if __name__ == "__main__":
    cmd = sys.argv[1] if len(sys.argv) > 1 else "emit"
    if cmd == "emit":
        j, c, skipped = generate()
        os.makedirs(os.path.dirname(OUT_C), exist_ok=True)
        open(OUT_JAVA, "w").write(j)
        open(OUT_C, "w").write(c)
        print(f"wrote {OUT_C} and {OUT_JAVA}; {len(skipped)} functions without a generated native: {skipped}")
    elif cmd == "collect":
        collect()
    elif cmd == "check":
        m, g, p = check()
        print(f"{m} names mapped to exported symbols, {g} explicit gaps, {len(p)} problems")
        for x in p:
            print("  ", x)
        sys.exit(1 if p else 0)
