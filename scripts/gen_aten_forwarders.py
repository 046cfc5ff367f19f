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
    "ATen.unique_dim": ("Tensor self, long dim, boolean sorted, boolean returnInverse, boolean returnCounts", "Tensor[]",
                        "return Tensor.owningAll(N.lamp_unique_dim(h(self), dim));"),
    "ATen.unique_consecutive": ("Tensor self, boolean returnInverse, boolean returnCounts, long dim", "Tensor[]",
                                "return Tensor.owningAll(N.lamp_unique_consecutive(h(self), dim));"),
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


DECL_RUNTIME_RE = re.compile(r"\b([a-z][A-Za-z0-9_]*)\s*:\s*(?:aten\.)?(STenOptions|TensorOptions|CudaStream|NcclComm)\b")


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


class Scopes:
    """Lexical scoping of a (comment-stripped) Scala source, as far as braces and parameter lists state it: which declaration `name: Type` a
    use of `name` at a given position refers to.  A declaration is visible at a position when the innermost `{ }` block around the declaration
    also contains the position; a declaration inside the parameter list of a `class` / `def` header belongs to the block (or, for a def
    without braces, the expression) that follows the header.  The NEAREST visible declaration wins - ops.scala declares `weight` as
    `Variable`, `Option[Variable]` and `STen` in neighbouring case classes, so file-wide types say nothing there."""

    def __init__(self, src):
        self.src = src
        n = len(src)
        self.block_of = [0] * (n + 1)            # position -> id of the innermost block (0 = file)
        self.blocks = [(0, n, 0)]                # id -> (start, end, parent)
        self.paren_open = [-1] * (n + 1)         # position -> start of the outermost open '(' / '[' inside the innermost block, or -1
        stack, parens, i = [0], [[]], 0
        while i < n:
            c = src[i]
            if c == '"':                         # string literals: no structure inside
                if src.startswith('"""', i):
                    j = src.find('"""', i + 3); j = n if j < 0 else j + 3
                else:
                    j = i + 1
                    while j < n and src[j] != '"' and src[j] != "\n":
                        j += 2 if src[j] == "\\" else 1
                    j = min(j + 1, n)
                for k in range(i, j):
                    self.block_of[k] = stack[-1]; self.paren_open[k] = parens[-1][0] if parens[-1] else -1
                i = j
                continue
            if c == "{":
                self.blocks.append((i, n, stack[-1])); stack.append(len(self.blocks) - 1); parens.append([])
            self.block_of[i] = stack[-1]
            self.paren_open[i] = parens[-1][0] if parens[-1] else -1
            if c in "([":
                parens[-1].append(i)
            elif c in ")]" and parens[-1]:
                parens[-1].pop()
            elif c == "}" and len(stack) > 1:
                b = stack.pop(); parens.pop()
                self.blocks[b] = (self.blocks[b][0], i, self.blocks[b][2])
            i += 1
        self.block_of[n] = 0
        self.decls = {}
        for m in DECL_RE.finditer(src):
            self.decls.setdefault(m.group(1), []).append((m.start(), m.group(2)))
        for m in DECL_RUNTIME_RE.finditer(src):
            self.decls.setdefault(m.group(1), []).append((m.start(), m.group(2)))
        # locals bound to a native's result have no written type: `val x = ATen.f(...)` is a Tensor, `val (a, b) = ATen.g(...)` are Tensors,
        # `val s = CudaStream.get...(...)` a CudaStream, `val c = NcclComm.comm_init_rank(...)` a NcclComm
        for m in re.finditer(r"\bval\s+([a-z][A-Za-z0-9_]*)\s*=\s*(?:aten\.)?(?:ATen|Tensor)\s*\.\s*[A-Za-z_0-9]+\s*\(", src):
            self.decls.setdefault(m.group(1), []).append((m.start(1), "Tensor"))
        for m in re.finditer(r"\bval\s*\(([^()=]*)\)\s*=\s*(?:aten\.)?ATen\s*\.", src):
            for nm in split_args(m.group(1)):
                if re.fullmatch(r"[a-z][A-Za-z0-9_]*", nm):
                    self.decls.setdefault(nm, []).append((m.start(1), "Tensor"))
        for m in re.finditer(r"\bval\s+([a-z][A-Za-z0-9_]*)\s*=\s*(?:aten\.)?CudaStream\s*\.\s*get", src):
            self.decls.setdefault(m.group(1), []).append((m.start(1), "CudaStream"))
        for m in re.finditer(r"\bval\s+([a-z][A-Za-z0-9_]*)\s*=\s*(?:aten\.)?NcclComm\s*\.\s*comm_init", src):
            self.decls.setdefault(m.group(1), []).append((m.start(1), "NcclComm"))
        for k in self.decls:
            self.decls[k].sort()

    def contains(self, block, pos):
        s, e, _ = self.blocks[block]
        return block == 0 or s < pos <= e

    def header_scope(self, p):
        """(start, end) of what a parameter declared at p (inside a header's parameter list) is visible in"""
        src, n = self.src, len(self.src)
        i, depth = self.paren_open[p], 0
        while i < n:                              # the end of this parameter list, and of the ones that follow it directly
            depth += src[i] in "(["; depth -= src[i] in ")]"
            i += 1
            if depth == 0:
                j = i
                while j < n and src[j] in " \n\t":
                    j += 1
                if j < n and src[j] == "(":
                    i = j; continue
                break
        # up to the body: `extends X(...)`, `: Type`, `=`; a `{` opens the body block, otherwise the expression runs to the next def / class / blank line
        m = re.compile(r"\{|=(?![=>])|\bdef\s|\bclass\s|\bobject\s|\n\s*\n").search(src, i)
        if m and m.group(0) == "=":                # a def's body: a block, or an expression that ends with the line its brackets close on
            j = m.end()
            while j < n and src[j] in " \n\t":
                j += 1
            if j < n and src[j] == "{":
                m = re.compile(r"\{").search(src, j)
            else:
                depth, k = 0, j
                while k < n:
                    depth += src[k] in "([{"; depth -= src[k] in ")]}"
                    if depth < 0:
                        break
                    if src[k] == "\n" and depth == 0 and not re.match(r"\s*\.", src[k + 1:k + 40]):
                        break
                    k += 1
                return i, k
        if m and m.group(0) == "{":
            b = self.block_of[m.start()]
            return self.blocks[b][0], self.blocks[b][1]
        return i, (m.start() if m else n)

    def type_of(self, name, pos):
        best = None
        for p, t in self.decls.get(name, ()):
            if p >= pos:
                break
            if self.paren_open[p] >= 0 and re.search(r"\b(class|def)\s+[^\s(\[=:{]+\s*(\[[^\]]*\])?\s*$", self.src[max(0, self.paren_open[p] - 120):self.paren_open[p]]):
                s, e = self.header_scope(p)
                ok = s < pos <= e
            else:
                ok = self.contains(self.block_of[p], pos)
            if ok:
                best = t
        return best


class SiteTypes:
    """the `types` argument of arg_kind for one call site: scoped declarations first, then what the file declares with one type only"""

    def __init__(self, scopes, pos, file_types):
        self.scopes, self.pos, self.file_types = scopes, pos, file_types

    def get(self, name, default=None):
        t = self.scopes.type_of(name, self.pos)
        return t if t is not None else self.file_types.get(name, default)

    def __contains__(self, name):
        return self.get(name) is not None

    def __getitem__(self, name):
        return self.get(name)


def block_value_tuple(src, scopes, call_start, with_names=False):
    """`val (a, b, c) = { ...; val r = ATen.x(...); ...; r }` (STen.scala:559-581): the native's result is bound to a name that is the VALUE of a
    block, and the block is destructured - the arity of that pattern, or 0"""
    none = (0, []) if with_names else 0
    m = re.search(r"\bval\s+([A-Za-z_][A-Za-z0-9_]*)\s*(?::[^=]+)?=\s*$", src[max(0, call_start - 120):call_start])
    if not m:
        return none
    b = scopes.block_of[call_start]
    if b == 0:
        return none
    s, e, _ = scopes.blocks[b]
    last = re.search(r"([A-Za-z_][A-Za-z0-9_]*)\s*$", src[s + 1:e])
    if not last or last.group(1) != m.group(1):
        return none
    tm = re.search(r"val\s*\(([^()=]*)\)\s*=\s*$", src[max(0, s - 200):s])
    k = len(split_args(tm.group(1))) if tm else 0
    if k < 2:
        return none
    return (k, split_args(tm.group(1))) if with_names else k



# ---- tuple ELEMENT types: a destructured native result is typed per element by what the Scala does with each name ----
def _balanced_end(src, i):
    """position just after the bracket that closes the one opened at src[i - 1]"""
    depth, n = 1, len(src)
    while i < n and depth:
        c = src[i]
        if c == '"':
            i += 1
            while i < n and src[i] != '"':
                i += 2 if src[i] == "\\" else 1
        elif c in "([{":
            depth += 1
        elif c in ")]}":
            depth -= 1
        i += 1
    return i


def _calls_in(text):
    """(callee, [argument texts], start) for every `name(`, `a.b.name(` or `name[T](` call in text (nested calls included)"""
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_.]*)\s*(?:\[[^\]\[]*\])?\s*\(", text):
        e = _balanced_end(text, m.end())
        yield m.group(1), split_args(text[m.end():e - 1]), m.start()


def _tensor_use(text, name):
    """the name is handled as an aten.Tensor: wrapped (`owned(x)`, `x.owned`), released, registered with a scope, or passed on to another native"""
    n = re.escape(name)
    if re.search(r"\bowned\(\s*%s\s*\)" % n, text) or re.search(r"\b%s\s*\.\s*(owned|release|value|sizes|numel|options|scalarTypeByte|to[A-Z])\b" % n, text):
        return True
    if re.search(r"\bregister\(\s*%s\s*\)" % n, text):
        return True
    for callee, args, _ in _calls_in(text):
        if re.match(r"(?:aten\.)?(ATen|Tensor)\.", callee) and any(re.fullmatch(r"(?:Option|Some)\(\s*%s\s*\)|%s" % (n, n), a) for a in args):
            return True
        if callee.endswith("releaseAll") and re.search(r"\b%s\b" % n, " ".join(args)):
            return True
    return False


def _bare_tuple_position(text, name):
    """index of `name` as an UNWRAPPED member of a tuple expression `( ..., name, ... )` (a parenthesis that is no call's argument list), or None"""
    for m in re.finditer(r"(?<![A-Za-z0-9_\]\)])\s*\(", text):
        if re.search(r"\bval\s*$", text[:m.start() + 1].rstrip("(").rstrip()):
            continue                                        # the destructuring pattern itself
        e = _balanced_end(text, m.end())
        parts = split_args(text[m.end():e - 1])
        if len(parts) >= 2 and name in parts:
            return parts.index(name), len(parts)
    return None


def _def_params(sources, fname):
    """parameter (name, type) lists of every `def fname` in the given sources"""
    out = []
    for src in sources.values():
        for m in re.finditer(r"\bdef\s+%s\b\s*(?:\[[^\]]*\])?\s*\(" % re.escape(fname), src):
            e = _balanced_end(src, m.end())
            ps = []
            for a in split_args(src[m.end():e - 1]):
                pm = re.match(r"\s*([A-Za-z_][A-Za-z0-9_]*)\s*:\s*(.+?)\s*(?:=.*)?$", a, re.S)
                if pm:
                    ps.append((pm.group(1), " ".join(pm.group(2).split())))
            out.append(ps)
    return out


def _kind_by_consumers(sources, defname, index, arity, depth=0):
    """a def returns the element unwrapped at `index` of an `arity`-tuple: follow the callers that destructure that result and read the element's
    type off the DECLARED type of the parameter it is handed to next (STen.scala:536-537 -> ops.scala:2352, 2375-2376 -> STen.scala:555-556)"""
    found = set()
    for src in sources.values():
        for m in re.finditer(r"\bval\s*\(([^()=]*)\)\s*=\s*(?:[A-Za-z_][A-Za-z0-9_]*\s*\.\s*)*%s\b\s*(?:\[[^\]]*\])?\s*\(" % re.escape(defname), src):
            names = split_args(m.group(1))
            if len(names) != arity or not re.fullmatch(r"[A-Za-z_][A-Za-z0-9_]*", names[index]) or names[index] == "_":
                continue
            nm = names[index]
            rest = src[m.end():]
            for callee, args, _ in _calls_in(rest):
                for j, a in enumerate(args):
                    named = re.fullmatch(r"([A-Za-z_][A-Za-z0-9_]*)\s*=\s*%s" % re.escape(nm), a)
                    if a != nm and not named:
                        continue
                    for ps in _def_params(sources, callee.split(".")[-1]):
                        if named:
                            ts = [t for n_, t in ps if n_ == named.group(1)]
                        else:
                            ts = [ps[j][1]] if j < len(ps) else []
                        for t in ts:
                            k = type_kind(t)
                            if k != "expr":
                                found.add(k)
    return found


def tuple_element_kinds(src, scopes, call_start, call_end, names, sources):
    """per destructured name: "tensor", a scalar kind ("long" / "double" / "bool") the element is declared as downstream, "unused", or "unknown"."""
    b = scopes.block_of[call_start]
    scope_end = scopes.blocks[b][1] if b else len(src)
    text = src[call_end:scope_end]
    dm = None
    for dm in re.finditer(r"\bdef\s+([A-Za-z_][A-Za-z0-9_]*)", src[:call_start]):
        pass
    kinds = []
    for nm in names:
        if nm == "_" or not re.search(r"\b%s\b" % re.escape(nm), text):
            kinds.append("unused"); continue
        if _tensor_use(text, nm):
            kinds.append("tensor"); continue
        pos = _bare_tuple_position(text, nm)
        got = set()
        if pos is not None and dm is not None:
            got = _kind_by_consumers(sources, dm.group(1), pos[0], pos[1])
        kinds.append(sorted(got)[0] if len(got) == 1 else "unknown")
    return kinds

def type_kind(t):
    if t == "Boolean":
        return "bool"
    if t in ("Int", "Long", "Short", "Byte"):
        return "long"
    if t in ("Double", "Float"):
        return "double"
    if t == "String":
        return "string"
    if t in ("STen", "Tensor", "Variable"):
        return "tensor"
    if t.startswith("Option["):
        inner = t[7:-1]
        return "option:" + {"STen": "tensor", "Tensor": "tensor", "Variable": "tensor"}.get(inner, type_kind(inner) if inner in ("Boolean", "Int", "Long", "Double", "Float", "String") else "expr")
    if re.match(r"Array\[Int\]", t):
        return "ints"
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
    om = re.match(r"([A-Za-z_][A-Za-z0-9_]*)\.map\(", a)
    if om:                                                   # the balanced end of `.map( ... )`, then nothing or `.getOrElse( ... )`
        i, depth = om.end(), 1
        while i < len(a) and depth:
            depth += a[i] in "([{"; depth -= a[i] in ")]}"; i += 1
        rest = a[i:]
        om = (om.group(1), a[om.end():i - 1], rest) if depth == 0 and (rest == "" or re.fullmatch(r"\.getOrElse\(.*\)", rest)) else None
    if om and str(types.get(om[0], "")).startswith("Option["):
        inner = type_kind(types.get(om[0]))                  # option:tensor for Option[STen | Tensor | Variable]
        if inner == "option:tensor" and not re.search(r"_(\.value)+\s*$|=>.*\.value\s*$", om[1]):
            inner = "option:expr"                            # mapped to something that is not the wrapped aten.Tensor
        return inner.split(":")[-1] if om[2] else inner
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
    am = re.fullmatch(r"Array\((.*)\)", a)
    if am:
        ek = {arg_kind(x, types) for x in split_args(am.group(1))}
        if "tensor" in ek and ek <= {"tensor", "expr"}:
            return "tensors"
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
    sources = {}
    for d in G.REF_DIRS:
        for f in sorted(glob.glob(os.path.join(ref, d, "**", "*.scala"), recursive=True)):
            sources[f] = strip_comments(open(f).read())
    for d in G.REF_DIRS:
        for f in sorted(glob.glob(os.path.join(ref, d, "**", "*.scala"), recursive=True)):
            src = sources[f]
            rel = os.path.relpath(f, ref)
            file_types = declared_types(src)
            scopes = Scopes(src)
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
                types = SiteTypes(scopes, m.start(), file_types)
                line = src.count("\n", 0, m.start()) + 1
                e = sites[cls].setdefault(name, {"calls": []})
                # `val (a, b, c) = ATen.x(...)`: the call's result is destructured as a tuple of that many members
                before = src[max(0, m.start() - 200):m.start()]
                tm = re.search(r"val\s*\(([^()=]*)\)\s*=\s*(?:[A-Za-z_.]*\(\s*)?$", before)
                tup = len(split_args(tm.group(1))) if tm else 0
                tup = tup if tup >= 2 else 0                       # `val (x) = ...` is no tuple
                names = split_args(tm.group(1)) if tup else []
                if not tup:
                    tup, names = block_value_tuple(src, scopes, m.start(), with_names=True)
                site = {"at": f"{rel}:{line}", "arity": len(args), "kinds": [arg_kind(a, types) for a in args], "tuple": tup}
                if tup:
                    # what each destructured member IS, by its downstream use (a tensor is wrapped / released / registered / passed to a native;
                    # a member returned unwrapped is typed by the declared parameter its consumer hands it to)
                    end = i
                    if not tm:                                 # block-valued: the names are used after the block
                        end = scopes.blocks[scopes.block_of[m.start()]][1] + 1
                    site["elements"] = tuple_element_kinds(src, scopes, m.start() if tm else end, end, names, sources)
                e["calls"].append(site)
    out = {"_source": {"dirs": G.REF_DIRS, "note": "call sites of the aten package in lamp's hot-path modules: argument counts, argument kinds (literals, Option / Some / None, arrays, tensors, booleans, identifiers by their declared type) and the arity of a destructured result - no source text "
                                                   "(scripts/gen_aten_forwarders.py collect, build container)"}}
    for c in CLASSES:
        out[c] = {n: sites[c][n] for n in sorted(sites[c])}
    json.dump(out, open(CALLSITES, "w"), indent=1)
    print({c: len(out[c]) for c in CLASSES}, sum(len(v["calls"]) for c in CLASSES for v in out[c].values()), "call sites")


INSTANCE_SITES = os.path.join(ROOT, "tests", "golden", "aten_instance_sites.json")
VALUE_STEP = {"Variable": "STen", "STen": "Tensor", "STenOptions": "TensorOptions"}       # what `.value` of each wrapper is
NOT_ATEN_MEMBERS = {"owned", "toLongArray", "toDoubleArray", "toFloatArray", "toMat", "toVec", "shape", "location", "getShape", "getCpu", "getScalarType",
                    "getStackTrace", "getBirth", "getValue"}                          # lamp's own extension methods / TensorTraceData's getters


def collect_instances():
    """what lamp calls on INSTANCES of aten.{Tensor, TensorOptions, CudaStream, NcclComm}: receiver chains `x`, `x.value`, `x.value.value` whose
    root the scoped declarations type as Variable / STen / Tensor / STenOptions / TensorOptions / CudaStream / NcclComm -> method, arity, kinds"""
    ref = "/root/reference"
    res = {}
    for d in G.REF_DIRS:
        for f in sorted(glob.glob(os.path.join(ref, d, "**", "*.scala"), recursive=True)):
            src = strip_comments(open(f).read())
            rel = os.path.relpath(f, ref)
            sc = Scopes(src)
            for m in re.finditer(r"(?<![A-Za-z0-9_.])([a-z][A-Za-z0-9_]*)((?:\s*\.\s*value)*)\s*\.\s*([A-Za-z_][A-Za-z0-9_]*)\b([ \t]*\()?", src):
                root, chain, meth, paren = m.group(1), m.group(2), m.group(3), m.group(4)
                if meth == "value" or meth in NOT_ATEN_MEMBERS:
                    continue
                t = sc.type_of(root, m.start())
                if t is None or t.startswith("Option["):
                    continue
                for _ in range(len(re.findall("value", chain))):
                    t = VALUE_STEP.get(t)
                    if t is None:
                        break
                if t not in ("Tensor", "TensorOptions", "CudaStream", "NcclComm"):
                    continue
                args = []
                if paren:
                    i, depth = m.end(), 1
                    while i < len(src) and depth:
                        depth += src[i] in "([{"; depth -= src[i] in ")]}"; i += 1
                    args = split_args(src[m.end():i - 1])
                line = src.count("\n", 0, m.start()) + 1
                site = {"at": f"{rel}:{line}", "arity": len(args), "kinds": [arg_kind(a, SiteTypes(sc, m.start(), {})) for a in args]}
                res.setdefault(t, {}).setdefault(meth, {"calls": []})["calls"].append(site)
    out = {"_source": {"dirs": G.REF_DIRS, "note": "instance-method call sites of the aten runtime classes in lamp's hot-path modules (receiver typed lexically through "
                                                   "`.value` chains): method, argument count, argument kinds - no source text (scripts/gen_aten_forwarders.py collect)"}}
    for c in sorted(res):
        out[c] = {n: res[c][n] for n in sorted(res[c])}
    json.dump(out, open(INSTANCE_SITES, "w"), indent=1)
    print({c: len(out[c]) for c in out if not c.startswith("_")}, "instance methods")


def instance_members():
    """(class, method) -> (result type, [parameter types]) of the PUBLIC NON-STATIC members the hand-written blocks define"""
    out = {}
    srcs = dict(SUPPORT)
    srcs["TensorOptions"] = TENSOR_OPTIONS
    for cls, text in srcs.items():
        for m in re.finditer(r"^  public (?!static)(\S+(?:<[^;{]*>)?(?:\[\])?) (\w+)\(([^)]*)\)", text, re.M):
            plist = [] if not m.group(3).strip() else [q.strip() for q in m.group(3).split(",")]
            out.setdefault((cls, m.group(2)), []).append((m.group(1), [" ".join(q.split()[:-1]) for q in plist]))
    return out


def check_instances():
    """every collected instance call site meets a member with that name, that many parameters and compatible kinds"""
    sites = json.load(open(INSTANCE_SITES))
    members = instance_members()
    problems = []
    for cls, table in sites.items():
        if cls.startswith("_"):
            continue
        for meth, e in table.items():
            cands = members.get((cls, meth))
            if not cands:
                problems.append(f"{cls}.{meth}: no member ({e['calls'][0]['at']})")
                continue
            for c in e["calls"]:
                if not any(len(pt) == c["arity"] and all(compatible(jt, k) for jt, k in zip(pt, c["kinds"])) for _, pt in cands):
                    problems.append(f"{cls}.{meth} at {c['at']}: passes {c['kinds']}, members take {[pt for _, pt in cands]}")
    return problems


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


ELEMENT_JAVA = {"tensor": "Tensor", "long": "Long", "double": "Double", "bool": "Boolean", "unused": "Tensor"}    # boxed: members of a scala.TupleN


def tuple_parts(e, k):
    """Java type of each member of the k-tuple the call sites destructure, from the collector's element kinds (`elements`); a member no site
    can type ("unknown") or that two sites type differently is an error - the table is not allowed to guess"""
    parts = []
    for i in range(k):
        ks = {c["elements"][i] for c in e["calls"] if c.get("tuple", 0) == k and "elements" in c} - {"unused"}
        assert len(ks) <= 1 and "unknown" not in ks, f"tuple member {i}: call sites say {sorted(ks)}"
        parts.append(ELEMENT_JAVA[next(iter(ks))] if ks else "Tensor")
    return parts


def compatible(jt, kind):
    if kind == "expr":
        return True
    if kind.startswith("option:"):
        return jt.startswith("scala.Option<")
    if kind == "array":
        return jt.endswith("[]")
    # (a Long argument where the forwarder takes a double: Scala widens it)
    return jt in {"tensor": ("Tensor",), "bool": ("boolean",), "long": ("long", "int", "short", "byte", "double"), "double": ("double", "float"),
                  "longs": ("long[]",), "ints": ("int[]",), "tensors": ("Tensor[]",), "bools": ("boolean[]",), "options": ("TensorOptions",),
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
            elif t:
                # member TYPES too: `val (.., max_q, ..) = ATen.f(...)` with max_q handed on to a `Long` parameter needs a Long there
                members = [x.strip() for x in rtype[rtype.index("<") + 1:rtype.rindex(">")].split(",")]
                for i, (jt, k) in enumerate(zip(members, c.get("elements", []))):
                    if k not in ("unused",) and ELEMENT_JAVA.get(k) != jt:
                        report["kind_mismatch"].append({"name": key, "at": c["at"], "position": f"result._{i + 1}", "call_site_passes": k, "forwarder_takes": jt})

    for cls in CLASSES:
        for name, e in sites.get(cls, {}).items():
            sym, _ = symbol_of(cls, name, nm, syms)
            key = f"{cls}.{name}"
            if name in HANDWRITTEN.get(cls, ()):
                hm = re.search(r"public static (\S+(?:<[^;{]*>)?(?:\[\])?) %s\(([^)]*)\)" % re.escape(name), SUPPORT[cls])
                assert hm, f"{key}: listed as hand-written but not in the support block"
                plist = [] if not hm.group(2).strip() else [q.strip() for q in hm.group(2).split(",")]
                arities = sorted({c["arity"] for c in e["calls"]})
                check_kinds(key, e, [" ".join(q.split()[:-1]) for q in plist], hm.group(1))
                report["forwarded"] += 1
                report.setdefault("handwritten", []).append(key)
                if arities != [len(plist)]:
                    report["arity_mismatch"].append({"name": key, "symbol": "hand-written", "forwarder_arity": len(plist), "call_site_arities": arities,
                                                     "at": [c["at"] for c in e["calls"] if c["arity"] != len(plist)][:3]})
                continue
            if key in EXPLICIT:
                params, rtype, body = EXPLICIT[key][:3]
                plist = [] if not params.strip() else [p.strip() for p in params.split(",")]
                n_params = len(plist)
                arities = sorted({c["arity"] for c in e["calls"]})
                where = e["calls"][0]["at"]
                tk = tuple_arity(e)
                if tk and rtype in ("Tensor[]", "Object[]"):
                    # the call sites destructure the result: the body (which builds an array) moves into a private method, the public one wraps
                    parts = tuple_parts(e, tk)
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
                assert tuple_parts(e, tk) == ["Tensor"] * tk, f"{cls}.{name}: the call sites use non-tensor members of a native that returns handles only"
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
    # Hand-written members of the aten classes: what lamp calls on INSTANCES (collected lexically into tests/golden/aten_instance_sites.json by
    # `collect`; tests/golden/aten_descriptors.json is the hand-audited table tests/test_jni_map.py holds these sources to).
    "Tensor": """  /** the C-ABI handle (lamp_tensor*); 0 after release() and for Tensor.undefined() */
  private long handle;
  private Tensor(long h) { handle = h; }
  private static final java.util.concurrent.ConcurrentHashMap<Long, StackTraceElement[]> births = new java.util.concurrent.ConcurrentHashMap<>();
  static volatile boolean tracing = false;
  static StackTraceElement[] birthOf(long h) { StackTraceElement[] t = births.get(h); return t == null ? new StackTraceElement[0] : t; }
  static Tensor owning(long h) {
    if (h == 0) return null;
    if (tracing) births.put(h, new Throwable().getStackTrace());
    return new Tensor(h);
  }
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
  /** STen.scala:559-580: the placeholder lamp passes for an absent optional tensor; release() on it is a no-op */
  public static Tensor undefined() { return new Tensor(0L); }
  public static boolean hasCuda() { return LampNative.lamp_has_gpu() != 0; }                       // STen.scala:1895 `if (aten.Tensor.hasCuda())`
  public static int getNumGPUs() { return LampNative.lamp_get_num_gpus(); }                        // device.scala:217
  public static void allowtf32(boolean flag) { LampNative.lamp_allow_tf32(flag ? 1 : 0); }
  public static void setPinnedMemoryAllocator() {}                                                  // pinned staging is lamp_pin_memory: nothing to install
  public void release() { if (handle != 0) { births.remove(handle); LampNative.lamp_tensor_release(handle); handle = 0; } }
  public static void releaseAll(Tensor[] ts) {
    for (Tensor t : ts) if (t != null) births.remove(t.handle);
    LampNative.lamp_tensor_release_all(handlesOf(ts));
    for (Tensor t : ts) if (t != null) t.handle = 0;
  }
  // ---- metadata (STen.scala:845-870) ----
  public long[] sizes() { return LampNative.lamp_tensor_sizes(handle); }
  public long[] strides() { return LampNative.lamp_tensor_strides(handle); }
  public long numel() { return LampNative.lamp_tensor_numel(handle); }
  public long dim() { return LampNative.lamp_tensor_ndim(handle); }
  public long elementSize() { return LampNative.lamp_tensor_element_size(handle); }
  public byte scalarTypeByte() { return (byte) LampNative.lamp_tensor_scalar_type(handle); }
  public boolean isCuda() { return LampNative.lamp_tensor_device(handle) >= 0; }
  public boolean is_pinned() { return LampNative.lamp_tensor_is_pinned(handle) != 0; }
  /** a NEW TensorOptions the caller releases (STen.scala:864, device.scala:221-225) */
  public TensorOptions options() { return new TensorOptions((byte) LampNative.lamp_tensor_scalar_type(handle), LampNative.lamp_tensor_device(handle)); }
  // ---- copies and placement (STen.scala:918-944, 1895-1898; device.scala:223) ----
  public Tensor to(TensorOptions options, boolean nonBlocking, boolean copy) {
    return owning(LampNative.lamp_to(handle, options.scalarTypeByte(), options.deviceIndex(), nonBlocking ? 1 : 0, copy ? 1 : 0));
  }
  public Tensor cpu() { return owning(LampNative.lamp_to(handle, LampNative.lamp_tensor_scalar_type(handle), -1, 0, 0)); }
  public Tensor pin_memory() { return owning(LampNative.lamp_pin_memory(handle)); }
  public void copyFrom(Tensor source, boolean nonBlocking) { LampNative.lamp_copy_(handle, handleOf(source), nonBlocking ? 1 : 0); }
  public Tensor expand_as(Tensor other) { return owning(LampNative.lamp_expand_as(handle, handleOf(other))); }
  /** Tensor.repeat (STen.scala:1758): the tensor tiled `repeats[d]` times along each dimension */
  public Tensor repeat(long[] repeats) {
    long[] sz = sizes();
    int lead = repeats.length - sz.length;
    if (lead < 0) throw new IllegalArgumentException("repeat: fewer repeats than dimensions");
    long[] v = new long[2 * repeats.length], e = new long[2 * repeats.length], o = new long[repeats.length];
    for (int d = 0; d < repeats.length; d++) {
      long s = d < lead ? 1 : sz[d - lead];
      v[2 * d] = 1; v[2 * d + 1] = s; e[2 * d] = repeats[d]; e[2 * d + 1] = s; o[d] = repeats[d] * s;
    }
    long a = LampNative.lamp_reshape(handle, v), b = LampNative.lamp_expand(a, e), c = LampNative.lamp_reshape(b, o);
    long r = LampNative.lamp_clone(c);
    LampNative.lamp_tensor_release_all(new long[] {a, b, c});
    return owning(r);
  }
  // ---- in-place scalar arithmetic (STen.scala:1123-1127, 1193-1197) ----
  public void add_(double other, double alpha) { LampNative.lamp_add_scalar_(handle, other, alpha); }
  public void add_l_(long other, long alpha) { LampNative.lamp_add_scalar_(handle, (double) other, (double) alpha); }
  public void mul_(double other) { LampNative.lamp_mul_scalar_(handle, other); }
  public void mul_l_(long other) { LampNative.lamp_mul_scalar_(handle, (double) other); }
  // ---- host arrays (TensorHelpers.scala:57-253): CPU tensors, `false` on failure as lamp asserts on the result ----
  public boolean copyFromDoubleArray(double[] a) { return LampNative.copyFromDoubleArray(handle, a); }
  public boolean copyFromFloatArray(float[] a) { return LampNative.copyFromFloatArray(handle, a); }
  public boolean copyFromLongArray(long[] a) { return LampNative.copyFromLongArray(handle, a); }
  public boolean copyFromIntArray(int[] a) { return LampNative.copyFromIntArray(handle, a); }
  public boolean copyFromShortArray(short[] a) { return LampNative.copyFromShortArray(handle, a); }
  public boolean copyFromByteArray(byte[] a) { return LampNative.copyFromByteArray(handle, a); }
  public boolean copyToDoubleArray(double[] a) { return LampNative.copyToDoubleArray(handle, a); }
  public boolean copyToFloatArray(float[] a) { return LampNative.copyToFloatArray(handle, a); }
  public boolean copyToLongArray(long[] a) { return LampNative.copyToLongArray(handle, a); }
  public boolean copyToIntArray(int[] a) { return LampNative.copyToIntArray(handle, a); }
  public boolean copyToShortArray(short[] a) { return LampNative.copyToShortArray(handle, a); }
  public boolean copyToByteArray(byte[] a) { return LampNative.copyToByteArray(handle, a); }
  /** the n elements from `offset` on of the flattened tensor (TensorHelpers.scala:190) */
  private long window(long offset, long n) {
    long f = LampNative.lamp_view(handle, new long[] {-1});
    try { return LampNative.lamp_narrow(f, 0, offset, n); } finally { LampNative.lamp_tensor_release(f); }
  }
  public boolean copyFromDoubleArrayAtOffset(double[] a, long offset) {
    try { long w = window(offset, a.length); try { return LampNative.copyFromDoubleArray(w, a); } finally { LampNative.lamp_tensor_release(w); } } catch (RuntimeException e) { return false; }
  }
  public boolean copyFromFloatArrayAtOffset(float[] a, long offset) {
    try { long w = window(offset, a.length); try { return LampNative.copyFromFloatArray(w, a); } finally { LampNative.lamp_tensor_release(w); } } catch (RuntimeException e) { return false; }
  }
  public boolean copyFromLongArrayAtOffset(long[] a, long offset) {
    try { long w = window(offset, a.length); try { return LampNative.copyFromLongArray(w, a); } finally { LampNative.lamp_tensor_release(w); } } catch (RuntimeException e) { return false; }
  }
  public boolean copyFromIntArrayAtOffset(int[] a, long offset) {
    try { long w = window(offset, a.length); try { return LampNative.copyFromIntArray(w, a); } finally { LampNative.lamp_tensor_release(w); } } catch (RuntimeException e) { return false; }
  }
  public boolean copyFromShortArrayAtOffset(short[] a, long offset) {
    try { long w = window(offset, a.length); try { return LampNative.copyFromShortArray(w, a); } finally { LampNative.lamp_tensor_release(w); } } catch (RuntimeException e) { return false; }
  }
  public boolean copyFromByteArrayAtOffset(byte[] a, long offset) {
    try { long w = window(offset, a.length); try { return LampNative.copyFromByteArray(w, a); } finally { LampNative.lamp_tensor_release(w); } } catch (RuntimeException e) { return false; }
  }
  // ---- sparse tensors: outside SURVEY section 8 (jni/name_map.json states the gap); named so that lamp-sten links ----
  public Tensor coalesce() { throw new UnsupportedOperationException("sparse tensors are not provided by liblamp_hip"); }
  public Tensor indices() { throw new UnsupportedOperationException("sparse tensors are not provided by liblamp_hip"); }
  public Tensor values() { throw new UnsupportedOperationException("sparse tensors are not provided by liblamp_hip"); }
  public Tensor to_dense() { throw new UnsupportedOperationException("sparse tensors are not provided by liblamp_hip"); }
""",
    "CudaStream": """  /** the C-ABI handle (lamp_stream*) */
  private final long handle;
  private CudaStream(long h) { handle = h; }
  static long handleOf(CudaStream s) { return s == null ? 0L : s.handle; }
  /** device.scala:181-208: `default.synchronize()`, `orig.synchronize()` */
  public void synchronize() { LampNative.lamp_stream_synchronize(handle); }
  /** the current stream and device are per OS thread (device.scala:119-129) */
  public static CudaStream getCurrentCUDAStream(byte device) { return new CudaStream(LampNative.lamp_stream_get_current(device)); }
  public static CudaStream getDefaultCUDAStream(byte device) { return new CudaStream(LampNative.lamp_stream_get_default(device)); }
  public static CudaStream getStreamFromPool(boolean highPriority, byte device) { return new CudaStream(LampNative.lamp_stream_get_from_pool(highPriority ? 1 : 0, device)); }
  public static void setCurrentCUDAStream(CudaStream s) { LampNative.lamp_stream_set_current(handleOf(s)); }
  public static int cudaGetDevice() { return LampNative.lamp_get_device(); }                       // STen.scala:636
  public static void cudaSetDevice(int device) { LampNative.lamp_set_device(device); }             // STen.scala:637, 639
""",
    "NcclComm": """  /** the C-ABI handle (lamp_comm*: an RCCL communicator) */
  private long handle;
  private NcclComm(long h) { handle = h; }
  static long[] handlesOf(NcclComm[] cs) { long[] r = new long[cs.length]; for (int i = 0; i < cs.length; i++) r[i] = cs[i] == null ? 0L : cs[i].handle; return r; }
  /** STen.scala:1906: `Base64.getEncoder.encodeToString(aten.NcclComm.get_unique_id)` */
  public static byte[] get_unique_id() { return LampNative.lamp_comm_get_unique_id(); }
  /** STen.scala:638: blocks until all ranks have joined */
  public static NcclComm comm_init_rank(int nRanks, byte[] uniqueId, int myRank) { return new NcclComm(LampNative.lamp_comm_init_rank(nRanks, uniqueId, myRank)); }
  /** STen.scala:648-651: one entry per GPU this thread drives (a group call); the root is rank 0 */
  public static void broadcast(Tensor[] tensors, NcclComm[] comms) { LampNative.lamp_comm_broadcast(Tensor.handlesOf(tensors), handlesOf(comms), 0); }
  /** STen.scala:664-670: op 0 = sum; the output lives on the root rank */
  public static void reduce(Tensor[] inputs, Tensor output, int rootRank, int op, NcclComm[] comms) {
    LampNative.lamp_comm_reduce(Tensor.handlesOf(inputs), Tensor.handleOf(output), rootRank, op, handlesOf(comms));
  }
  /** not in aten-scala: the one-call gradient exchange SURVEY section 8b asks the new backend to add */
  public static void all_reduce(Tensor[] tensors, NcclComm[] comms, int op) { LampNative.lamp_comm_all_reduce(Tensor.handlesOf(tensors), handlesOf(comms), op); }
  public void comm_destroy() { if (handle != 0) { LampNative.lamp_comm_destroy(handle); handle = 0; } }
""",
    "TensorTrace": """  private TensorTrace() {}
  /** TensorLogger.scala:200, 233 */
  public static void enable() { Tensor.tracing = true; LampNative.lamp_tensor_trace_enable(1); }
  public static void disable() { Tensor.tracing = false; LampNative.lamp_tensor_trace_enable(0); }
  /** TensorLogger.scala:23: `aten.TensorTrace.list.map(v => v.getValue)` - (handle, data) pairs of the live traced tensors.
   *  lamp_tensor_trace_list records are 14 longs: handle, birth (ns), scalar type, device, ndim, 8 sizes, bytes. */
  @SuppressWarnings("unchecked")
  public static java.util.Map.Entry<Long, TensorTraceData>[] list() {
    long[] rec = LampNative.lamp_tensor_trace_list();
    int n = rec.length / 14;
    java.util.Map.Entry<Long, TensorTraceData>[] out = (java.util.Map.Entry<Long, TensorTraceData>[]) new java.util.Map.Entry[n];
    for (int i = 0; i < n; i++) {
      int b = 14 * i, nd = (int) rec[b + 4];
      long[] shape = new long[nd];
      for (int d = 0; d < nd; d++) shape[d] = rec[b + 5 + d];
      out[i] = new java.util.AbstractMap.SimpleImmutableEntry<>(rec[b], new TensorTraceData(shape, rec[b + 3] < 0, (byte) rec[b + 2], rec[b + 1], Tensor.birthOf(rec[b])));
    }
    return out;
  }
""",
}

# classes without forwarders: written whole
EXTRA_CLASSES = {
    "TensorTraceData": """// GENERATED by scripts/gen_aten_forwarders.py - do not edit.
// aten.TensorTraceData as lamp.TensorLogger reads it (TensorLogger.scala:13-62): shape, device class, scalar type byte, birth time, creating stack.
package aten;

public final class TensorTraceData {
  private final long[] shape;
  private final boolean cpu;
  private final byte scalarType;
  private final long birth;
  private final StackTraceElement[] stackTrace;
  TensorTraceData(long[] shape, boolean cpu, byte scalarType, long birth, StackTraceElement[] stackTrace) {
    this.shape = shape; this.cpu = cpu; this.scalarType = scalarType; this.birth = birth; this.stackTrace = stackTrace;
  }
  public long[] getShape() { return shape; }
  public boolean getCpu() { return cpu; }
  public byte getScalarType() { return scalarType; }
  public long getBirth() { return birth; }
  public StackTraceElement[] getStackTrace() { return stackTrace; }
}
""",
    "TensorOptionsTrace": """// GENERATED by scripts/gen_aten_forwarders.py - do not edit.
// aten.TensorOptionsTrace (TensorLogger.scala:31, 201, 234): TensorOptions are plain JVM values here - no native object, nothing to leak or list.
package aten;

public final class TensorOptionsTrace {
  private TensorOptionsTrace() {}
  public static void enable() {}
  public static void disable() {}
  @SuppressWarnings("unchecked")
  public static java.util.Map.Entry<Long, TensorTraceData>[] list() { return (java.util.Map.Entry<Long, TensorTraceData>[]) new java.util.Map.Entry[0]; }
}
""",
}

# statics that the hand-written blocks above define: no generated forwarder beside them
HANDWRITTEN = {"Tensor": {"hasCuda", "getNumGPUs", "releaseAll", "undefined", "allowtf32", "setPinnedMemoryAllocator"},
               "CudaStream": {"getCurrentCUDAStream", "getDefaultCUDAStream", "getStreamFromPool", "setCurrentCUDAStream", "cudaGetDevice", "cudaSetDevice"},
               "NcclComm": {"get_unique_id", "comm_init_rank", "broadcast", "reduce"},
               "TensorTrace": {"enable", "disable", "list"}}


TENSOR_OPTIONS = '''// GENERATED by scripts/gen_aten_forwarders.py - do not edit.
// aten.TensorOptions as lamp uses it (STenOptions, lamp-sten/src/main/scala/lamp/STen.scala:678-780; statics STen.scala:18-35): a (scalar type,
// device) pair; the C ABI takes the two as ints (scalar type byte as in ATen: 0 u8, 1 i8, 2 i16, 3 i32, 4 i64, 5 f16, 6 f32, 7 f64, 11 bool,
// 15 bf16; device -1 = CPU, >= 0 = GPU ordinal).  Every method returns a NEW value; release() exists because lamp's Scope calls it.
package aten;

public final class TensorOptions {
  private final byte scalarType;
  private final int device;
  TensorOptions(byte scalarType, int device) { this.scalarType = scalarType; this.device = device; }
  public static TensorOptions dtypeDouble() { return new TensorOptions((byte) 7, -1); }
  public static TensorOptions dtypeFloat() { return new TensorOptions((byte) 6, -1); }
  public static TensorOptions dtypeLong() { return new TensorOptions((byte) 4, -1); }
  public static TensorOptions dtypeHalf() { return new TensorOptions((byte) 5, -1); }
  public static TensorOptions dtypeBF16() { return new TensorOptions((byte) 15, -1); }
  public static TensorOptions d() { return dtypeDouble(); }
  public static TensorOptions f() { return dtypeFloat(); }
  public static TensorOptions l() { return dtypeLong(); }
  public static TensorOptions i() { return new TensorOptions((byte) 3, -1); }
  public static TensorOptions sh() { return new TensorOptions((byte) 2, -1); }
  public static TensorOptions b() { return new TensorOptions((byte) 1, -1); }               // "compatible with Scala's Byte": signed 8 bit
  public static TensorOptions fromScalarType(byte scalarType) { return new TensorOptions(scalarType, -1); }
  public TensorOptions cpu() { return new TensorOptions(scalarType, -1); }
  public TensorOptions cuda() { return new TensorOptions(scalarType, 0); }
  public TensorOptions cuda_index(short index) { return new TensorOptions(scalarType, index); }
  /** device(deviceType, index): 0 cpu, 1 cuda (= the GPU here), 13 mps (STen.scala:757-759; no such device behind this library) */
  public TensorOptions device(byte deviceType, int index) {
    if (deviceType == 0) return new TensorOptions(scalarType, -1);
    if (deviceType == 1) return new TensorOptions(scalarType, index);
    throw new UnsupportedOperationException("TensorOptions.device: device type " + deviceType + " is not provided by liblamp_hip");
  }
  public TensorOptions toDouble() { return new TensorOptions((byte) 7, device); }
  public TensorOptions toFloat() { return new TensorOptions((byte) 6, device); }
  public TensorOptions toHalf() { return new TensorOptions((byte) 5, device); }
  public TensorOptions toBF16() { return new TensorOptions((byte) 15, device); }
  public TensorOptions toLong() { return new TensorOptions((byte) 4, device); }
  public TensorOptions toInt() { return new TensorOptions((byte) 3, device); }
  public TensorOptions toShort() { return new TensorOptions((byte) 2, device); }
  public TensorOptions toByte() { return new TensorOptions((byte) 1, device); }
  public boolean isDouble() { return scalarType == 7; }
  public boolean isFloat() { return scalarType == 6; }
  public boolean isLong() { return scalarType == 4; }
  public boolean isInt() { return scalarType == 3; }
  public boolean isShort() { return scalarType == 2; }
  public boolean isByte() { return scalarType == 1; }
  public boolean isCPU() { return device < 0; }
  public boolean isCuda() { return device >= 0; }
  public boolean isSparse() { return false; }
  public int deviceIndex() { return device; }
  public byte scalarTypeByte() { return scalarType; }
  public void release() {}
}
'''


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
    open(os.path.join(OUT_DIR, "TensorOptions.java"), "w").write(TENSOR_OPTIONS)
    for cls, text in EXTRA_CLASSES.items():
        open(os.path.join(OUT_DIR, f"{cls}.java"), "w").write(text)
    json.dump(report, open(os.path.join(OUT_DIR, "forwarders_report.json"), "w"), indent=1)
    print({k: (len(v) if isinstance(v, list) else v) for k, v in report.items()})


def check():
    _, report = forwarders()
    return report


if __name__ == "__main__":
    cmd = sys.argv[1] if len(sys.argv) > 1 else "emit"
    if cmd == "collect":
        collect()
        collect_instances()
    elif cmd == "emit":
        emit()
    else:
        r = check()
        print(json.dumps({k: (len(v) if isinstance(v, list) else v) for k, v in r.items()}))
        for m in r["arity_mismatch"] + r["kind_mismatch"]:
            print("  ", m)
        inst = check_instances()
        for m in inst:
            print("  ", m)
        sys.exit(1 if (r["arity_mismatch"] or r["kind_mismatch"] or inst) else 0)
