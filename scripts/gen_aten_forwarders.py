#!/usr/bin/env python3
"""The drop-in JVM classes of the `aten` package: jni/aten/{ATen,Tensor,TensorOptions,CudaStream,NcclComm,TensorTrace}.java.

lamp calls its backend as `aten.ATen.add_0(a, b, alpha)`, `aten.Tensor.addmm_out_transposed1(...)`, `aten.NcclComm.broadcast(...)` -
static methods with aten-scala's overload-suffixed names (aten-scala-core is not in the reference tree: build.sbt:125).  jni/LampNative.java
(scripts/gen_jni.py) exposes the C ABI one native per function; the classes written here are the forwarders with aten's names.

  python scripts/gen_aten_forwarders.py collect   (build container: needs /root/reference) parse every `ATen.x(...)` / `Tensor.x(...)` /
                                                  `CudaStream.x(...)` / `NcclComm.x(...)` / `TensorTrace.x(...)` call of lamp's hot-path modules:
                                                  argument count and literal kinds per call site -> tests/golden/aten_callsites.json (data only)
  python scripts/gen_aten_forwarders.py emit      forwarders for every collected name that jni/name_map.json maps to an exported symbol
  python scripts/gen_aten_forwarders.py check     every call site's arity agrees with the forwarder generated for its name (tests/test_jni_map.py)

How a forwarder's parameter list is derived: the C-ABI function's inputs in declaration order (scripts/gen_jni.py `plan`) ARE aten's
argument order - the header was written against the reference's call sites - with two systematic differences that are undone here:
  * aten passes ONE `TensorOptions` where the C ABI takes (dtype, device) ints;
  * aten passes tensors as `aten.Tensor` objects (a `long` handle inside), optional tensors as `java.util.Optional`-free nullable references.
Names whose call sites do not have the arity this derivation gives are listed in jni/name_map.json with an explicit `"args"` entry
(a list of expressions over the aten parameters p0, p1, ...), or reported by `check`."""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import gen_jni as G  # noqa: E402

CALLSITES = os.path.join(ROOT, "tests", "golden", "aten_callsites.json")
OUT_DIR = os.path.join(ROOT, "jni", "aten")
CLASSES = ("ATen", "Tensor", "CudaStream", "NcclComm", "TensorTrace")


# Names whose aten signature (from the call sites) is NOT the C-ABI function's input list: the native is a composition, takes fewer
# arguments (aten passes options the backend has one value for), or is the non-`out` form.  (aten parameter list, result type, Java body).
# N = LampNative, h = Tensor.handleOf, own = Tensor.owning; Scala Option arguments arrive as Object (Tensor.handleOfOption).
EXPLICIT = {
    "ATen._unique": ("Tensor self, boolean sorted, boolean returnInverse", "Tensor[]",
                     "long[] r = N.lamp_unique(h(self));\n    N.lamp_tensor_release(r[2]);\n    return new Tensor[] {own(r[0]), own(r[1])};"),
    "ATen._unique2": ("Tensor self, boolean sorted, boolean returnInverse, boolean returnCounts", "Tensor[]",
                      "return Tensor.owningAll(N.lamp_unique(h(self)));"),
    "ATen.index_copy_out": ("Tensor out, Tensor self, long dim, Tensor index, Tensor source", "void",
                            "long r = N.lamp_index_copy(h(self), dim, h(index), h(source));\n    N.lamp_copy_(h(out), r, 0);\n    N.lamp_tensor_release(r);"),
    "ATen.median_0": ("Tensor self", "Tensor",
                      "long f = N.lamp_view(h(self), new long[] {-1});\n    long[] r = N.lamp_median_dim(f, 0, 0);\n    N.lamp_tensor_release_all(new long[] {f, r[1]});\n    return own(r[0]);"),
    "ATen.scatter_1": ("Tensor self, long dim, Tensor index, double value", "Tensor",
                       "return own(N.lamp_scatter_value(h(self), dim, h(index), value));"),
    "ATen._log_softmax_backward_data": ("Tensor gradOutput, Tensor output, long dim, byte inputDtype", "Tensor",
                                        "return own(N.lamp_log_softmax_backward_data(h(gradOutput), h(output), dim));"),
    "ATen._scaled_dot_product_cudnn_attention": ("Tensor query, Tensor key, Tensor value, scala.Option<Tensor> attnBias, boolean computeLogSumExp, double dropoutP, boolean isCausal, boolean returnDebugMask", "Object[]",
        "long b = Tensor.handleOfOption(attnBias);\n    long[] r = b == 0 ? N.lamp_scaled_dot_product_attention(h(query), h(key), h(value), isCausal ? 1 : 0, 0.0)\n"
        "                      : N.lamp_scaled_dot_product_attention_bias(h(query), h(key), h(value), b, isCausal ? 1 : 0, 0.0);\n"
        "    // (output, logsumexp, cum_seq_q, cum_seq_k, max_q, max_k, philox_seed, philox_offset, debug_attn_mask): no dropout, so the bookkeeping tensors are empty\n"
        "    long[] z = {0};\n"
        "    return new Object[] {own(r[0]), own(r[1]), own(N.lamp_zeros(z, 4, -1)), own(N.lamp_zeros(z, 4, -1)), 0L, 0L, own(N.lamp_zeros(z, 4, -1)), own(N.lamp_zeros(z, 4, -1)), own(N.lamp_zeros(z, 6, -1))};"),
    "ATen._scaled_dot_product_cudnn_attention_backward": ("Tensor gradOutput, Tensor query, Tensor key, Tensor value, Tensor out, Tensor logsumexp, Tensor philoxSeed, Tensor philoxOffset, Tensor attnBias, Tensor cumSeqQ, Tensor cumSeqK, long maxQ, long maxK, double dropoutP, boolean isCausal", "Tensor[]",
        "long b = h(attnBias);\n    boolean defined = b != 0 && N.lamp_tensor_numel(b) > 0;\n"
        "    return Tensor.owningAll(defined ? N.lamp_scaled_dot_product_attention_bias_backward(h(gradOutput), h(query), h(key), h(value), h(out), h(logsumexp), b, isCausal ? 1 : 0, 0.0)\n"
        "                                    : N.lamp_scaled_dot_product_attention_backward(h(gradOutput), h(query), h(key), h(value), h(out), h(logsumexp), isCausal ? 1 : 0, 0.0));"),
    "ATen.addcmul": ("Tensor self, Tensor tensor1, Tensor tensor2, double value", "Tensor",
                     "long o = N.lamp_clone(h(self));\n    N.lamp_addcmul_out(o, h(self), h(tensor1), h(tensor2), value);\n    return own(o);"),
    "ATen.all_0": ("Tensor self, long dim, boolean keepDim", "Tensor",
                   "long nz = N.lamp_ne_scalar(h(self), 0.0), f = N.lamp_cast(nz, 6);\n    long cnt = N.lamp_sum_dims(f, new long[] {dim}, keepDim ? 1 : 0);\n"
                   "    long r = N.lamp_eq_scalar(cnt, (double) N.lamp_tensor_sizes(h(self))[(int) dim]);\n    N.lamp_tensor_release_all(new long[] {nz, f, cnt});\n    return own(r);"),
    "ATen.all_1": ("Tensor self", "Tensor",
                   "long nz = N.lamp_ne_scalar(h(self), 0.0), f = N.lamp_cast(nz, 6), cnt = N.lamp_sum_all(f);\n"
                   "    long r = N.lamp_eq_scalar(cnt, (double) N.lamp_tensor_numel(h(self)));\n    N.lamp_tensor_release_all(new long[] {nz, f, cnt});\n    return own(r);"),
    "ATen.any_0": ("Tensor self, long dim, boolean keepDim", "Tensor",
                   "long nz = N.lamp_ne_scalar(h(self), 0.0), f = N.lamp_cast(nz, 6);\n    long cnt = N.lamp_sum_dims(f, new long[] {dim}, keepDim ? 1 : 0);\n"
                   "    long r = N.lamp_gt_scalar(cnt, 0.0);\n    N.lamp_tensor_release_all(new long[] {nz, f, cnt});\n    return own(r);"),
    "ATen.any_1": ("Tensor self", "Tensor",
                   "long nz = N.lamp_ne_scalar(h(self), 0.0), f = N.lamp_cast(nz, 6), cnt = N.lamp_sum_all(f);\n"
                   "    long r = N.lamp_gt_scalar(cnt, 0.0);\n    N.lamp_tensor_release_all(new long[] {nz, f, cnt});\n    return own(r);"),
    "ATen.argmin": ("Tensor self, long dim, boolean keepDim", "Tensor",
                    "long n = N.lamp_neg(h(self));\n    long r = N.lamp_argmax(n, dim, keepDim ? 1 : 0);\n    N.lamp_tensor_release(n);\n    return own(r);"),
    "ATen.avg_pool2d": ("Tensor self, long[] kernelSize, long[] stride, long[] padding, boolean ceilMode, boolean countIncludePad, long divisorOverride", "Tensor",
                        "return own(N.lamp_avg_pool2d(h(self), kernelSize[0], stride[0], padding[0], ceilMode ? 1 : 0, countIncludePad ? 1 : 0));"),
    "ATen.avg_pool2d_backward": ("Tensor gradOutput, Tensor self, long[] kernelSize, long[] stride, long[] padding, boolean ceilMode, boolean countIncludePad, long divisorOverride", "Tensor",
                                 "return own(N.lamp_avg_pool2d_backward(h(gradOutput), h(self), kernelSize[0], stride[0], padding[0], ceilMode ? 1 : 0, countIncludePad ? 1 : 0));"),
    "ATen.binary_cross_entropy_with_logits": ("Tensor self, Tensor target, scala.Option<Tensor> weight, scala.Option<Tensor> posWeight, long reduction", "Tensor",
                                              "if (Tensor.handleOfOption(weight) != 0) throw new UnsupportedOperationException(\"binary_cross_entropy_with_logits: per-element weights are not used by lamp\");\n"
                                              "    return own(N.lamp_binary_cross_entropy_with_logits(h(self), h(target), Tensor.handleOfOption(posWeight), reduction));"),
    "ATen.conv1d_0": ("Tensor input, Tensor weight, Object bias, long[] stride, long[] padding, long[] dilation, long groups", "Tensor",
                      "return own(N.lamp_convolution(h(input), h(weight), Tensor.handleOfOption(bias), stride, padding, dilation, 0, new long[] {0}, groups));"),
    "ATen.conv_transpose1d": ("Tensor input, Tensor weight, Object bias, long[] stride, long[] padding, long[] outputPadding, long groups, long[] dilation", "Tensor",
                              "return own(N.lamp_convolution(h(input), h(weight), Tensor.handleOfOption(bias), stride, padding, dilation, 1, outputPadding, groups));"),
    "ATen.convolution_backward": ("Tensor gradOutput, Tensor input, Tensor weight, scala.Option<long[]> biasSizes, long[] stride, long[] padding, long[] dilation, boolean transposed, long[] outputPadding, long groups, boolean[] outputMask", "Tensor[]",
                                  "return Tensor.owningAll(N.lamp_convolution_backward(h(gradOutput), h(input), h(weight), stride, padding, dilation, transposed ? 1 : 0, outputPadding, groups, outputMask));"),
    "ATen.embedding": ("Tensor weight, Tensor indices, long paddingIdx, boolean scaleGradByFreq, boolean sparse", "Tensor",
                       "return own(N.lamp_embedding(h(weight), h(indices)));"),
    "ATen.embedding_backward": ("Tensor grad, Tensor indices, long numWeights, long paddingIdx, boolean scaleGradByFreq, boolean sparse", "Tensor",
                                "return own(N.lamp_embedding_backward(h(grad), h(indices), numWeights, paddingIdx));"),
    "ATen.eye_0": ("long n, TensorOptions options", "Tensor", "return own(N.lamp_eye(n, n, options.scalarTypeByte(), options.deviceIndex()));"),
    "ATen.gather": ("Tensor self, long dim, Tensor index, boolean sparseGrad", "Tensor", "return own(N.lamp_gather(h(self), dim, h(index)));"),
    "ATen.index": ("Tensor self, Tensor[] indices", "Tensor",
                   "if (indices.length != 1) throw new UnsupportedOperationException(\"ATen.index: lamp indexes with one index tensor along dimension 0\");\n"
                   "    return own(N.lamp_index_select(h(self), 0, h(indices[0])));"),
    "ATen.index_select_out": ("Tensor out, Tensor self, long dim, Tensor index", "void",
                              "long r = N.lamp_index_select(h(self), dim, h(index));\n    N.lamp_copy_(h(out), r, 0);\n    N.lamp_tensor_release(r);"),
    "ATen.max_0": ("Tensor self, long dim, boolean keepDim", "Tensor[]",
                   "long idx = N.lamp_argmax(h(self), dim, 1);\n    long val = N.lamp_gather(h(self), dim, idx);\n"
                   "    if (!keepDim) { long v2 = N.lamp_squeeze(val, dim), i2 = N.lamp_squeeze(idx, dim); N.lamp_tensor_release_all(new long[] {val, idx}); val = v2; idx = i2; }\n"
                   "    return new Tensor[] {own(val), own(idx)};"),
    "ATen.max_2": ("Tensor self, Tensor other", "Tensor", "return own(N.lamp_maximum(h(self), h(other)));"),
    "ATen.min_0": ("Tensor self, long dim, boolean keepDim", "Tensor[]",
                   "long neg = N.lamp_neg(h(self));\n    long idx = N.lamp_argmax(neg, dim, 1);\n    long val = N.lamp_gather(h(self), dim, idx);\n    N.lamp_tensor_release(neg);\n"
                   "    if (!keepDim) { long v2 = N.lamp_squeeze(val, dim), i2 = N.lamp_squeeze(idx, dim); N.lamp_tensor_release_all(new long[] {val, idx}); val = v2; idx = i2; }\n"
                   "    return new Tensor[] {own(val), own(idx)};"),
    "ATen.mean_1": ("Tensor self, long[] dim, boolean keepDim", "Tensor", "return own(N.lamp_mean_dims(h(self), dim, keepDim ? 1 : 0));"),
    "ATen.mean_out": ("Tensor out, Tensor self, long[] dim, boolean keepDim", "void",
                      "long r = N.lamp_mean_dims(h(self), dim, keepDim ? 1 : 0);\n    N.lamp_copy_(h(out), r, 0);\n    N.lamp_tensor_release(r);"),
    "ATen.nan_to_num": ("Tensor self", "Tensor", "return own(N.lamp_nan_to_num(h(self), 0.0));"),
    "ATen.norm_2": ("Tensor self, double p, long[] dim, boolean keepDim, byte dtype", "Tensor",
                    "if (p != 2.0) throw new UnsupportedOperationException(\"ATen.norm: lamp uses the 2-norm only\");\n    return own(N.lamp_norm2_dims(h(self), dim, keepDim ? 1 : 0));"),
    "ATen.norm_3": ("Tensor self, double p, long[] dim, boolean keepDim", "Tensor",
                    "if (p != 2.0) throw new UnsupportedOperationException(\"ATen.norm: lamp uses the 2-norm only\");\n    return own(N.lamp_norm2_dims(h(self), dim, keepDim ? 1 : 0));"),
    "ATen.pow_out_0": ("Tensor out, Tensor self, Tensor exponent", "void",
                       "long r = N.lamp_pow_tensor(h(self), h(exponent));\n    N.lamp_copy_(h(out), r, 0);\n    N.lamp_tensor_release(r);"),
    "ATen.pow_out_2": ("Tensor out, Tensor self, double exponent", "void",
                       "long r = N.lamp_pow_scalar(h(self), exponent);\n    N.lamp_copy_(h(out), r, 0);\n    N.lamp_tensor_release(r);"),
    "ATen.randint_0": ("long high, long[] size, TensorOptions options", "Tensor",
                       "return own(N.lamp_randint(0, high, size, options.scalarTypeByte(), options.deviceIndex()));"),
    "ATen.remainder_out_0": ("Tensor out, Tensor self, double other", "void",
                             "long r = N.lamp_remainder_scalar(h(self), other);\n    N.lamp_copy_(h(out), r, 0);\n    N.lamp_tensor_release(r);"),
    "ATen.remainder_out_1": ("Tensor out, Tensor self, Tensor other", "void",
                             "long r = N.lamp_remainder(h(self), h(other));\n    N.lamp_copy_(h(out), r, 0);\n    N.lamp_tensor_release(r);"),
    "ATen.repeat_interleave_0": ("Tensor repeats", "Tensor",
                                 "long n = N.lamp_tensor_numel(h(repeats));\n    long ar = N.lamp_arange(0.0, (double) n, 1.0, 4, N.lamp_tensor_device(h(repeats)));\n"
                                 "    long r = N.lamp_repeat_interleave_tensor(ar, h(repeats), 0);\n    N.lamp_tensor_release(ar);\n    return own(r);"),
    "ATen.sign_out": ("Tensor out, Tensor self", "void", "long r = N.lamp_sign(h(self));\n    N.lamp_copy_(h(out), r, 0);\n    N.lamp_tensor_release(r);"),
    "ATen.squeeze_0": ("Tensor self", "Tensor", "return own(N.lamp_squeeze(h(self), Long.MIN_VALUE));"),
    "ATen.std_0": ("Tensor self, boolean unbiased", "Tensor",
                   "long[] vm = N.lamp_var_mean_dims(h(self), new long[0], unbiased ? 1 : 0, 0);\n    long r = N.lamp_sqrt(vm[0]);\n    N.lamp_tensor_release_all(vm);\n    return own(r);"),
    "ATen.std_mean_0": ("Tensor self, boolean unbiased", "Tensor[]",
                        "long[] vm = N.lamp_var_mean_dims(h(self), new long[0], unbiased ? 1 : 0, 0);\n    long sd = N.lamp_sqrt(vm[0]);\n    N.lamp_tensor_release(vm[0]);\n    return new Tensor[] {own(sd), own(vm[1])};"),
    "ATen.sum_1": ("Tensor self, long[] dim, boolean keepDim", "Tensor", "return own(N.lamp_sum_dims(h(self), dim, keepDim ? 1 : 0));"),
    "ATen.sum_out": ("Tensor out, Tensor self, long[] dim, boolean keepDim", "void",
                     "long r = N.lamp_sum_dims(h(self), dim, keepDim ? 1 : 0);\n    N.lamp_copy_(h(out), r, 0);\n    N.lamp_tensor_release(r);"),
    "ATen.tensordot": ("Tensor self, Tensor other, long[] dimsSelf, long[] dimsOther", "Tensor",
                       "if (dimsSelf.length != 1 || dimsOther.length != 1 || N.lamp_tensor_ndim(h(self)) != 2 || N.lamp_tensor_ndim(h(other)) != 2)\n"
                       "      throw new UnsupportedOperationException(\"ATen.tensordot: matrices contracted over one dimension\");\n"
                       "    long a = dimsSelf[0] == 1 ? N.lamp_tensor_retain(h(self)) : N.lamp_transpose(h(self), 0, 1);\n"
                       "    long b = dimsOther[0] == 0 ? N.lamp_tensor_retain(h(other)) : N.lamp_transpose(h(other), 0, 1);\n"
                       "    long r = N.lamp_mm(a, b);\n    N.lamp_tensor_release_all(new long[] {a, b});\n    return own(r);"),
    "ATen.var_0": ("Tensor self, boolean unbiased", "Tensor",
                   "long[] vm = N.lamp_var_mean_dims(h(self), new long[0], unbiased ? 1 : 0, 0);\n    N.lamp_tensor_release(vm[1]);\n    return own(vm[0]);"),
    "ATen.var_mean_0": ("Tensor self, boolean unbiased", "Tensor[]", "return Tensor.owningAll(N.lamp_var_mean_dims(h(self), new long[0], unbiased ? 1 : 0, 0));"),
    "ATen.where_4": ("Tensor condition", "Tensor[]",
                     "throw new UnsupportedOperationException(\"ATen.where(condition) (= nonzero as a tuple): outside lamp's hot path, not provided by liblamp_hip\");"),
    "ATen.zeros_like": ("Tensor self, TensorOptions options", "Tensor",
                        "return own(N.lamp_zeros(N.lamp_tensor_sizes(h(self)), options.scalarTypeByte(), options.deviceIndex()));"),
    "Tensor.manual_seed_cuda": ("long seed, int device", "void", "N.lamp_manual_seed(seed);"),
    "Tensor.from_file": ("String path, long offset, long length, byte scalarType, boolean pin", "Tensor",
                         "return own(N.lamp_tensors_from_file(path, offset, length, pin, new long[] {scalarType}, new long[] {0}, new long[] {length})[0]);"),
    "Tensor.tensors_from_file": ("String path, long offset, long length, boolean pin, byte[] scalarTypes, long[] tensorOffsets, long[] tensorLengths", "Tensor[]",
                                 "long[] types = new long[scalarTypes.length];\n    for (int i = 0; i < types.length; i++) types[i] = scalarTypes[i];\n"
                                 "    return owningAll(N.lamp_tensors_from_file(path, offset, length, pin, types, tensorOffsets, tensorLengths));"),
    "NcclComm.comm_init_rank": ("int nranks, byte[] uniqueId, int rank", "long", "return N.lamp_comm_init_rank(nranks, uniqueId, rank);"),
    "NcclComm.broadcast": ("Tensor[] tensors, long[] comms", "void", "N.lamp_comm_broadcast(Tensor.handlesOf(tensors), comms, 0);"),
    "TensorTrace.disable": ("", "void", "N.lamp_tensor_trace_enable(0);"),
    "TensorTrace.enable": ("", "void", "N.lamp_tensor_trace_enable(1);"),
    # --- same arity as the native, different meaning of an argument (found by comparing literal kinds at the call sites) ---
    "ATen._cast_Char": ("Tensor self, boolean nonBlocking", "Tensor", "return own(N.lamp_cast(h(self), 1));"),
    "ATen._cast_Short": ("Tensor self, boolean nonBlocking", "Tensor", "return own(N.lamp_cast(h(self), 2));"),
    "ATen._cast_Int": ("Tensor self, boolean nonBlocking", "Tensor", "return own(N.lamp_cast(h(self), 3));"),
    "ATen._cast_Long": ("Tensor self, boolean nonBlocking", "Tensor", "return own(N.lamp_cast(h(self), 4));"),
    "ATen._cast_Half": ("Tensor self, boolean nonBlocking", "Tensor", "return own(N.lamp_cast(h(self), 5));"),
    "ATen._cast_Float": ("Tensor self, boolean nonBlocking", "Tensor", "return own(N.lamp_cast(h(self), 6));"),
    "ATen._cast_Double": ("Tensor self, boolean nonBlocking", "Tensor", "return own(N.lamp_cast(h(self), 7));"),
    "ATen.fill__1": ("Tensor self, Tensor value", "void", "N.lamp_fill_(h(self), N.lamp_item(h(value)));"),
    "ATen.index_fill_1": ("Tensor self, long dim, Tensor index, Tensor value", "Tensor", "return own(N.lamp_index_fill(h(self), dim, h(index), N.lamp_item(h(value))));"),
    "ATen.max_pool1d_with_indices": ("Tensor self, long[] kernelSize, long[] stride, long[] padding, long[] dilation, boolean ceilMode", "Tensor[]",
                                     "return Tensor.owningAll(N.lamp_max_pool1d_with_indices(h(self), kernelSize[0], stride[0], padding[0], dilation[0], ceilMode ? 1 : 0));"),
    "ATen.max_pool2d_with_indices": ("Tensor self, long[] kernelSize, long[] stride, long[] padding, long[] dilation, boolean ceilMode", "Tensor[]",
                                     "return Tensor.owningAll(N.lamp_max_pool2d_with_indices(h(self), kernelSize[0], stride[0], padding[0], dilation[0], ceilMode ? 1 : 0));"),
    "ATen.max_pool2d_with_indices_backward": ("Tensor gradOutput, Tensor self, long[] kernelSize, long[] stride, long[] padding, long[] dilation, boolean ceilMode, Tensor indices", "Tensor",
                                              "return own(N.lamp_max_pool2d_with_indices_backward(h(gradOutput), h(self), kernelSize[0], stride[0], padding[0], dilation[0], ceilMode ? 1 : 0, h(indices)));"),
    "ATen.mul_1": ("Tensor self, double other", "Tensor", "return own(N.lamp_mul_scalar(h(self), other));"),
    "ATen.narrow_1": ("Tensor self, long dim, Tensor start, long length", "Tensor", "return own(N.lamp_narrow(h(self), dim, (long) N.lamp_item(h(start)), length));"),
    "ATen.pow_0": ("Tensor self, Tensor exponent", "Tensor", "return own(N.lamp_pow_tensor(h(self), h(exponent)));"),
    "ATen.repeat_interleave_1": ("Tensor self, Tensor repeats, long dim", "Tensor", "return own(N.lamp_repeat_interleave_tensor(h(self), h(repeats), dim));"),
}


def split_args(s):
    """top-level comma split of a Scala argument list (balanced (), [], {}; string literals)"""
    out, depth, cur, i = [], 0, "", 0
    while i < len(s):
        ch = s[i]
        if ch == '"':
            j = i + 1
            while j < len(s) and s[j] != '"':
                j += 2 if s[j] == "\\" else 1
            cur += s[i:j + 1]; i = j + 1
            continue
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
        i += 1
    if cur.strip():
        out.append(cur.strip())
    return out


DECL_RE = re.compile(r"\b([a-z][A-Za-z0-9_]*)\s*:\s*(Boolean|Int|Long|Short|Byte|Double|Float|String|Option\[[A-Za-z\[\]]+\]|(?:Seq|List|Array|Vector)\[[A-Za-z\[\]]+\]|STen|Tensor|Variable)")


def strip_comments(src):
    """comments blanked out (newlines kept, so line numbers stay): commented-out calls are not call sites"""
    out, i, n = [], 0, len(src)
    while i < n:
        c = src[i]
        if c == '"':
            if src.startswith('"""', i):
                j = src.find('"""', i + 3)
                j = n if j < 0 else j + 3
            else:
                j = i + 1
                while j < n and src[j] != '"' and src[j] != "\n":
                    j += 2 if src[j] == "\\" else 1
                j = min(j + 1, n)
            out.append(src[i:j]); i = j
        elif src.startswith("//", i):
            j = src.find("\n", i)
            j = n if j < 0 else j
            out.append(" " * (j - i)); i = j
        elif src.startswith("/*", i):
            j = src.find("*/", i + 2)
            j = n if j < 0 else j + 2
            out.append("".join(ch if ch == "\n" else " " for ch in src[i:j])); i = j
        else:
            out.append(c); i += 1
    return "".join(out)


def declared_types(src):
    """identifier -> declared Scala type, for the identifiers a file declares with ONE type (parameters, vals, fields): what a call site's
    argument is can then be read off lexically"""
    seen = {}
    for m in DECL_RE.finditer(src):
        seen.setdefault(m.group(1), set()).add(m.group(2))
    return {k: next(iter(v)) for k, v in seen.items() if len(v) == 1}


def type_kind(t):
    if t == "Boolean":
        return "bool"
    if t in ("Int", "Long", "Short", "Byte"):
        return "long"
    if t in ("Double", "Float"):
        return "double"
    if t == "String":
        return "string"
    if t.startswith("Option["):
        inner = t[7:-1]
        return "option:" + {"STen": "tensor", "Tensor": "tensor", "Variable": "tensor"}.get(inner, type_kind(inner) if inner in ("Boolean", "Int", "Long", "Double", "Float", "String") else "expr")
    if re.match(r"(Seq|List|Array|Vector)\[(Int|Long)\]", t):
        return "longs"
    if re.match(r"(Seq|List|Array|Vector)\[(STen|Tensor|Variable)\]", t):
        return "tensors"
    return "expr"


def arg_kind(a, types=None):
    """what a call site passes at one position, as far as the text says: literal kinds, Option(...) / Some(...) / None, arrays, `.value` tensors,
    boolean expressions, and identifiers whose declared type the file states"""
    types = types or {}
    a = re.sub(r"\s+", " ", a).strip()
    if a in ("true", "false"):
        return "bool"
    if re.fullmatch(r"-?\d+[lL]?", a):
        return "long"
    if re.fullmatch(r"-?\d*\.\d+(e-?\d+)?[dD]?|-?\d+[dD]|-?\d+e-?\d+", a):
        return "double"
    if a == "None":
        return "option:none"
    m = re.fullmatch(r"(?:Option|Some)\((.*)\)", a)
    if m:
        return "option:" + arg_kind(m.group(1), types).split(":")[-1]
    if re.search(r"\.map\(\s*_\.value\s*\)$", a) and not re.search(r"\.toArray", a):
        base = re.sub(r"\.map\(\s*_\.value\s*\)$", "", a)
        t = types.get(base.split(".")[-1], "")
        return "tensors" if re.match(r"(Seq|List|Array|Vector)\[", t) else "option:tensor"
    if re.search(r"(?i)(options|opt)[A-Za-z0-9]*\b[^,]*\.value$|\.options(\([^)]*\))?(\.value)?$|TensorOptions\.", a):
        return "options"
    if re.search(r"map\(_(\._\d)?\.value\)\.toArray|\.toArray\.map\(_\.value\)|Array\(.*value", a):
        return "tensors"
    if re.fullmatch(r"Array\((\s*(true|false)\s*,?)+\)", a):
        return "bools"
    if re.search(r"Array\(|Array\.|map\(_\.toLong\)", a):
        return "longs"
    if re.search(r"\.toArray", a):
        base = re.sub(r"\.toArray.*$", "", a)
        t = types.get(base.split(".")[-1], "") if re.fullmatch(r"[A-Za-z_][A-Za-z0-9_.]*", base) else ""
        k = type_kind(t) if t else "expr"
        return k if k in ("longs", "tensors") else "array"
    if a.endswith(".value") or a.endswith(".value)"):
        return "tensor"
    if a == "value":
        return type_kind(types["value"]) if "value" in types else "tensor"
    if a.startswith('"'):
        return "string"
    if re.search(r"\.toLong$|\.toInt$", a):
        return "long"
    if re.search(r"\.toDouble$|\.toFloat$", a):
        return "double"
    if re.search(r"==|!=|<=|>=|&&|\|\||^!|\.isDefined$|\.isEmpty$|\.nonEmpty$", a):
        return "bool"
    if re.fullmatch(r"[A-Za-z_][A-Za-z0-9_]*(\.[A-Za-z_][A-Za-z0-9_]*)*", a):
        t = types.get(a.split(".")[-1])
        if t:
            return type_kind(t)
    return "expr"


def collect():
    ref = "/root/reference"
    sites = {c: {} for c in CLASSES}
    for d in G.REF_DIRS:
        for f in sorted(glob.glob(os.path.join(ref, d, "**", "*.scala"), recursive=True)):
            src = strip_comments(open(f).read())
            rel = os.path.relpath(f, ref)
            file_types = declared_types(src)
            defs = [(d.start(), d.end()) for d in re.finditer(r"\bdef\s+[^\s(\[=:]+", src)]
            for m in re.finditer(r"\b(ATen|Tensor|CudaStream|NcclComm|TensorTrace)\s*\.\s*([A-Za-z_][A-Za-z_0-9]*)\s*\(", src):
                cls, name = m.group(1), m.group(2)
                i, depth = m.end(), 1
                while i < len(src) and depth:
                    if src[i] == '"':
                        i += 1
                        while i < len(src) and src[i] != '"':
                            i += 2 if src[i] == "\\" else 1
                    elif src[i] in "([{":
                        depth += 1
                    elif src[i] in ")]}":
                        depth -= 1
                    i += 1
                args = split_args(src[m.end():i - 1])
                # identifiers are typed by the parameter list of the enclosing def first (STen.scala declares `other` as STen, Double and Long in
                # neighbouring overloads), then by what the file declares unambiguously
                types = dict(file_types)
                enclosing = [d for d in defs if d[0] < m.start()]
                if enclosing:
                    head = src[enclosing[-1][1]:m.start()]
                    cut = re.search(r"\)\s*(:\s*[A-Za-z\[\], ().]+)?\s*=\s", head)
                    sig_text = head[:cut.start() + 1] if cut else head[:400]
                    for dm in DECL_RE.finditer(sig_text):
                        types[dm.group(1)] = dm.group(2)
                line = src.count("\n", 0, m.start()) + 1
                e = sites[cls].setdefault(name, {"calls": []})
                # `val (a, b, c) = ATen.x(...)`: the call's result is destructured as a tuple of that many members
                before = src[max(0, m.start() - 200):m.start()]
                tm = re.search(r"val\s*\(([^()=]*)\)\s*=\s*(?:[A-Za-z_.]*\(\s*)?$", before)
                tup = len(split_args(tm.group(1))) if tm else 0
                tup = tup if tup >= 2 else 0                       # `val (x) = ...` is no tuple
                e["calls"].append({"at": f"{rel}:{line}", "arity": len(args), "kinds": [arg_kind(a, types) for a in args], "tuple": tup})
    out = {"_source": {"dirs": G.REF_DIRS, "note": "call sites of the aten package in lamp's hot-path modules: argument counts, argument kinds (literals, Option / Some / None, arrays, tensors, booleans, identifiers by their declared type) and the arity of a destructured result - no source text "
                                                   "(scripts/gen_aten_forwarders.py collect, build container)"}}
    for c in CLASSES:
        out[c] = {n: sites[c][n] for n in sorted(sites[c])}
    json.dump(out, open(CALLSITES, "w"), indent=1)
    print({c: len(out[c]) for c in CLASSES}, sum(len(v["calls"]) for c in CLASSES for v in out[c].values()), "call sites")


def native_signatures():
    sig = {}
    for name, ret, plist in G.parse_header():
        if ret != "int" or name in G.MANUAL:
            continue
        pl = G.plan(name, plist)
        if pl is None:
            continue
        ins, outs = pl
        jret, _ = G.result_types(outs)
        if jret is None:
            continue
        sig[name] = (ins, outs, jret)
    return sig


def symbol_of(cls, name, nm, syms):
    table = nm.get(cls, {})
    e = table.get(name) or table.get(G.base_name(name))
    if e is None:
        guess = "lamp_" + G.base_name(name).lstrip("_")
        return (guess if (cls == "ATen" and guess in syms) else None), None
    if "symbol" not in e:
        return None, None
    s = e["symbol"] if isinstance(e["symbol"], str) else e["symbol"][0]
    return (None if s.startswith("jvm:") else s), e.get("args")


def aten_params(ins):
    """the aten-side parameter list implied by a native's inputs: (java type, name, expression handed to the native)"""
    ps, i = [], 0
    while i < len(ins):
        p = ins[i]
        nxt = ins[i + 1] if i + 1 < len(ins) else None
        if p.kind == "int" and p.cname in ("dtype", "scalar_type") and nxt is not None and nxt.kind == "int" and nxt.cname.startswith("device"):
            ps.append(("TensorOptions", "options", ["options.scalarTypeByte()", "options.deviceIndex()"]))
            i += 2
            continue
        if p.kind == "handle":
            ps.append(("Tensor", p.cname, [f"Tensor.handleOf({p.cname})"]))
        elif p.kind in ("handles", "handles+count"):
            ps.append(("Tensor[]", p.cname, [f"Tensor.handlesOf({p.cname})"]))
        elif p.kind in ("longs", "longs+count"):
            ps.append(("long[]", p.cname, [p.cname]))
        elif p.kind == "doubles":
            ps.append(("double[]", p.cname, [p.cname]))
        elif p.kind == "mask":
            ps.append(("boolean[]", p.cname, [p.cname]))
        elif p.kind == "string":
            ps.append(("String", p.cname, [p.cname]))
        elif p.kind == "int":
            # aten's flags are booleans, its enum-like ints are longs: both arrive as a long here (boolean call sites pass through Flag.of)
            ps.append(("long", p.cname, [f"(int) {p.cname}"]))
        elif p.kind == "long":
            ps.append(("long", p.cname, [p.cname]))
        else:
            ps.append(("double", p.cname, [p.cname]))
        i += 1
    return ps


def wrap_result(jret, outs):
    if jret == "void":
        return "void", "{call};"
    if jret == "long" and len(outs) == 1 and outs[0].kind == "handle":
        return "Tensor", "return Tensor.owning({call});"
    if jret == "long[]" and all(o.kind in ("handle", "handle_array") for o in outs):
        return "Tensor[]", "return Tensor.owningAll({call});"
    return jret, "return {call};"


def kinds_at(e, n_params):
    """per parameter position: the kinds the call sites (of that arity) pass there"""
    ks = [set() for _ in range(n_params)]
    for c in e["calls"]:
        if c["arity"] == n_params:
            for i, k in enumerate(c["kinds"]):
                ks[i].add(k)
    return ks


def tuple_arity(e):
    t = {c.get("tuple", 0) for c in e["calls"]} - {0}
    return max(t) if t else 0


def compatible(jt, kind):
    if kind == "expr":
        return True
    if kind.startswith("option:"):
        return jt.startswith("scala.Option<")
    if kind == "array":
        return jt.endswith("[]")
    # (a Long argument where the forwarder takes a double: Scala widens it)
    return jt in {"tensor": ("Tensor",), "bool": ("boolean",), "long": ("long", "int", "short", "byte", "double"), "double": ("double", "float"),
                  "longs": ("long[]",), "tensors": ("Tensor[]",), "bools": ("boolean[]",), "options": ("TensorOptions",),
                  "string": ("String",)}.get(kind, (jt,))


def refine_params(ps, kinds):
    """the native's input list re-typed by what the call sites pass: scala.Option where they pass Option / Some / None, boolean where they pass
    Booleans (the C ABI takes flags as ints)"""
    out = []
    for (jt, name, exprs), ks in zip(ps, kinds):
        if any(k.startswith("option:") for k in ks):
            if jt == "Tensor":
                jt, exprs = "scala.Option<Tensor>", [f"Tensor.handleOfOption({name})"]
            elif jt == "long[]":
                jt, exprs = "scala.Option<long[]>", [f"Tensor.longsOfOption({name})"]
        elif "bool" in ks and jt == "long" and exprs == [f"(int) {name}"]:
            jt, exprs = "boolean", [f"{name} ? 1 : 0"]
        out.append((jt, name, exprs))
    return out


def tuple_type(k, parts):
    return f"scala.Tuple{k}<{', '.join(parts)}>"


def forwarders():
    sites = json.load(open(CALLSITES))
    nm = json.load(open(G.NAME_MAP))
    syms = G.exported_symbols()
    sig = native_signatures()
    per_class, report = {c: [] for c in CLASSES}, {"forwarded": 0, "arity_mismatch": [], "kind_mismatch": [], "unmapped": [], "no_native": []}

    def check_kinds(key, e, jtypes, rtype):
        for c in e["calls"]:
            if c["arity"] != len(jtypes):
                continue
            for i, (jt, k) in enumerate(zip(jtypes, c["kinds"])):
                if not compatible(jt, k):
                    report["kind_mismatch"].append({"name": key, "at": c["at"], "position": i, "call_site_passes": k, "forwarder_takes": jt})
            t = c.get("tuple", 0)
            if t and not rtype.startswith(f"scala.Tuple{t}<"):
                report["kind_mismatch"].append({"name": key, "at": c["at"], "position": "result", "call_site_passes": f"val ({t} names) =", "forwarder_takes": rtype})

    for cls in CLASSES:
        for name, e in sites.get(cls, {}).items():
            sym, _ = symbol_of(cls, name, nm, syms)
            key = f"{cls}.{name}"
            if key in EXPLICIT:
                params, rtype, body = EXPLICIT[key][:3]
                plist = [] if not params.strip() else [p.strip() for p in params.split(",")]
                n_params = len(plist)
                arities = sorted({c["arity"] for c in e["calls"]})
                where = e["calls"][0]["at"]
                tk = tuple_arity(e)
                if tk and rtype in ("Tensor[]", "Object[]"):
                    # the call sites destructure the result: the body (which builds an array) moves into a private method, the public one wraps
                    parts = list(EXPLICIT[key][3]) if len(EXPLICIT[key]) > 3 else ["Tensor"] * tk
                    tt = tuple_type(tk, parts)
                    args_ = ", ".join(p.split()[-1] for p in plist)
                    items = ", ".join(f"({parts[i]}) r_[{i}]" for i in range(tk))
                    text = (f"  /** {cls}.{name} ({where}) */\n  public static {tt} {name}({params}) {{\n    {rtype} r_ = {name}__parts({args_});\n"
                            f"    return new {tt}({items});\n  }}\n  private static {rtype} {name}__parts({params}) {{\n    {body}\n  }}")
                    rtype = tt
                else:
                    text = f"  /** {cls}.{name} ({where}) */\n  public static {rtype} {name}({params}) {{\n    {body}\n  }}"
                per_class[cls].append((name, text, n_params, arities))
                check_kinds(key, e, [" ".join(p.split()[:-1]) for p in plist], rtype)
                report["forwarded"] += 1
                report.setdefault("explicit", []).append(key)
                if arities != [n_params]:
                    report["arity_mismatch"].append({"name": key, "symbol": "explicit", "forwarder_arity": n_params, "call_site_arities": arities,
                                                     "at": [c["at"] for c in e["calls"] if c["arity"] != n_params][:3]})
                continue
            if sym is None:
                report["unmapped"].append(f"{cls}.{name}")
                continue
            if sym not in sig:
                if key != "Tensor.releaseAll":              # hand-written in the support block of aten.Tensor
                    report["no_native"].append(f"{cls}.{name} -> {sym}")
                continue
            ins, outs, jret = sig[sym]
            ps = aten_params(ins)
            ps = refine_params(ps, kinds_at(e, len(ps)))
            arities = sorted({c["arity"] for c in e["calls"]})
            rtype, body = wrap_result(jret, outs)
            tk = tuple_arity(e)
            if tk and rtype == "Tensor[]" and len([o for o in outs]) >= 1:
                tt = tuple_type(tk, ["Tensor"] * tk)
                rtype, body = tt, "long[] r_ = {call}; return new " + tt + "(" + ", ".join(f"Tensor.owning(r_[{i}])" for i in range(tk)) + ");"
            call = f"LampNative.{sym}({', '.join(x for _, _, xs in ps for x in xs)})"
            decl = ", ".join(f"{t} {n}" for t, n, _ in ps)
            where = e["calls"][0]["at"]
            per_class[cls].append((name, f"  /** {cls}.{name} - {sym} ({where}) */\n  public static {rtype} {name}({decl}) {{ {body.format(call=call)} }}", len(ps), arities))
            check_kinds(f"{cls}.{name}", e, [t for t, _, _ in ps], rtype)
            report["forwarded"] += 1
            if arities != [len(ps)]:
                report["arity_mismatch"].append({"name": f"{cls}.{name}", "symbol": sym, "forwarder_arity": len(ps), "call_site_arities": arities,
                                                 "at": [c["at"] for c in e["calls"] if c["arity"] != len(ps)][:3]})
    return per_class, report


SUPPORT = {
    "Tensor": '''  /** the C-ABI handle (lamp_tensor*); 0 after release() */
  private long handle;
  private Tensor(long h) { handle = h; }
  static Tensor owning(long h) { return h == 0 ? null : new Tensor(h); }
  static Tensor[] owningAll(long[] hs) { Tensor[] r = new Tensor[hs.length]; for (int i = 0; i < hs.length; i++) r[i] = owning(hs[i]); return r; }
  static long handleOf(Tensor t) { return t == null ? 0L : t.handle; }
  /** a Scala Option[Tensor] (or a Tensor, or null): lamp passes optional tensors as scala.Option */
  static long handleOfOption(Object o) {
    if (o == null) return 0L;
    if (o instanceof Tensor) return ((Tensor) o).handle;
    try {
      if ((Boolean) o.getClass().getMethod("isEmpty").invoke(o)) return 0L;
      return handleOf((Tensor) o.getClass().getMethod("get").invoke(o));
    } catch (ReflectiveOperationException e) { throw new IllegalArgumentException("expected scala.Option[aten.Tensor], got " + o.getClass(), e); }
  }
  /** a Scala Option[Array[Long]] (or a long[], or null) */
  static long[] longsOfOption(Object o) {
    if (o == null) return null;
    if (o instanceof long[]) return (long[]) o;
    try {
      if ((Boolean) o.getClass().getMethod("isEmpty").invoke(o)) return null;
      return (long[]) o.getClass().getMethod("get").invoke(o);
    } catch (ReflectiveOperationException e) { throw new IllegalArgumentException("expected scala.Option[Array[Long]], got " + o.getClass(), e); }
  }
  static long[] handlesOf(Tensor[] ts) { long[] r = new long[ts.length]; for (int i = 0; i < ts.length; i++) r[i] = handleOf(ts[i]); return r; }
  public void release() { if (handle != 0) { LampNative.lamp_tensor_release(handle); handle = 0; } }
  public static void releaseAll(Tensor[] ts) { LampNative.lamp_tensor_release_all(handlesOf(ts)); for (Tensor t : ts) if (t != null) t.handle = 0; }
  public long[] sizes() { return LampNative.lamp_tensor_sizes(handle); }
  public long[] strides() { return LampNative.lamp_tensor_strides(handle); }
  public long numel() { return LampNative.lamp_tensor_numel(handle); }
  public long dim() { return LampNative.lamp_tensor_ndim(handle); }
  public byte scalarTypeByte() { return (byte) LampNative.lamp_tensor_scalar_type(handle); }
  public TensorOptions options() { return new TensorOptions((byte) LampNative.lamp_tensor_scalar_type(handle), LampNative.lamp_tensor_device(handle)); }
  public boolean copyFromDoubleArray(double[] a) { return LampNative.copyFromDoubleArray(handle, a); }
  public boolean copyFromFloatArray(float[] a) { return LampNative.copyFromFloatArray(handle, a); }
  public boolean copyFromLongArray(long[] a) { return LampNative.copyFromLongArray(handle, a); }
  public boolean copyToDoubleArray(double[] a) { return LampNative.copyToDoubleArray(handle, a); }
  public boolean copyToFloatArray(float[] a) { return LampNative.copyToFloatArray(handle, a); }
  public boolean copyToLongArray(long[] a) { return LampNative.copyToLongArray(handle, a); }
''',
}


HELPERS = """  private static final class N extends LampNative {}     // (static natives: `N.lamp_x(...)` reads shorter)
  private static long h(Tensor t) { return Tensor.handleOf(t); }
  private static Tensor own(long handle) { return Tensor.owning(handle); }
"""


def emit():
    per_class, report = forwarders()
    os.makedirs(OUT_DIR, exist_ok=True)
    for cls in CLASSES:
        body = "\n".join(t for _, t, _, _ in sorted(per_class[cls]))
        src = (f"// GENERATED by scripts/gen_aten_forwarders.py from include/lamp_hip.h, jni/name_map.json and tests/golden/aten_callsites.json - do not edit.\n"
               f"// aten.{cls}: the static methods lamp calls on this class, forwarded to the natives of aten.LampNative (jni/aten_jni.c over liblamp_hip.so).\n"
               f"package aten;\n\npublic final class {cls} {{\n" + SUPPORT.get(cls, f"  private {cls}() {{}}\n") + HELPERS + "\n" + body + "\n}\n")
        open(os.path.join(OUT_DIR, f"{cls}.java"), "w").write(src)
    open(os.path.join(OUT_DIR, "TensorOptions.java"), "w").write('''// GENERATED by scripts/gen_aten_forwarders.py - do not edit.
// aten.TensorOptions as lamp uses it (STenOptions, lamp-sten/src/main/scala/lamp/STen.scala): a (scalar type, device) pair; the C ABI takes the
// two as ints (scalar type byte as in ATen: 0 u8, 2 i16, 3 i32, 4 i64, 5 f16, 6 f32, 7 f64, 11 bool, 15 bf16; device -1 = CPU, >= 0 = GPU ordinal).
package aten;

public final class TensorOptions {
  private final byte scalarType;
  private final int device;
  TensorOptions(byte scalarType, int device) { this.scalarType = scalarType; this.device = device; }
  public static TensorOptions dtypeDouble() { return new TensorOptions((byte) 7, -1); }
  public static TensorOptions dtypeFloat() { return new TensorOptions((byte) 6, -1); }
  public static TensorOptions dtypeLong() { return new TensorOptions((byte) 4, -1); }
  public static TensorOptions dtypeHalf() { return new TensorOptions((byte) 5, -1); }
  public static TensorOptions dtypeBFloat16() { return new TensorOptions((byte) 15, -1); }
  public static TensorOptions d() { return dtypeDouble(); }
  public static TensorOptions f() { return dtypeFloat(); }
  public static TensorOptions l() { return dtypeLong(); }
  public TensorOptions cpu() { return new TensorOptions(scalarType, -1); }
  public TensorOptions cuda() { return new TensorOptions(scalarType, 0); }
  public TensorOptions cuda_index(short i) { return new TensorOptions(scalarType, i); }
  public TensorOptions toDouble() { return new TensorOptions((byte) 7, device); }
  public TensorOptions toFloat() { return new TensorOptions((byte) 6, device); }
  public TensorOptions toLong() { return new TensorOptions((byte) 4, device); }
  public boolean isCPU() { return device < 0; }
  public boolean isCuda() { return device >= 0; }
  public int deviceIndex() { return device; }
  public byte scalarTypeByte() { return scalarType; }
  public void release() {}
}
''')
    json.dump(report, open(os.path.join(OUT_DIR, "forwarders_report.json"), "w"), indent=1)
    print({k: (len(v) if isinstance(v, list) else v) for k, v in report.items()})


def check():
    _, report = forwarders()
    return report


if __name__ == "__main__":
    cmd = sys.argv[1] if len(sys.argv) > 1 else "emit"
    if cmd == "collect":
        collect()
    elif cmd == "emit":
        emit()
    else:
        r = check()
        print(json.dumps({k: (len(v) if isinstance(v, list) else v) for k, v in r.items()}))
        for m in r["arity_mismatch"] + r["kind_mismatch"]:
            print("  ", m)
        sys.exit(1 if (r["arity_mismatch"] or r["kind_mismatch"]) else 0)
