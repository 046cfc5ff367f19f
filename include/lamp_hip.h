/*
 * lamp_hip.h - C ABI of liblamp_hip.so, the MI355X (gfx950) native tensor backend for Lamp.
 *
 * This is the drop-in boundary: the functions below are what a JNI adapter for the JVM
 * package `aten` (the un-vendored dependency io.github.pityka:aten-scala-core, reference
 * build.sbt:125) would bind, one native per `aten.ATen.*` / `aten.Tensor.*` /
 * `aten.CudaStream.*` / `aten.NcclComm.*` call that lamp's hot path makes.  Each group cites
 * the reference call sites (paths relative to the reference root) it replaces.  See
 * INTEGRATION.md for the JNI stub a maintainer would add.
 *
 * Conventions (reference: how lamp uses the handles)
 *  - every function returns 0 on success, non-zero on failure; lamp_last_error() then holds a
 *    thread-local message (libtorch raises c10::Error -> JVM exception; the JNI shim does
 *    the same from this status).  Shape/dtype/device mismatches fail loudly, never corrupt.
 *  - tensors are opaque handles.  Every tensor returned through an out-parameter is a NEW
 *    handle the caller owns (also for views, lamp releases views separately:
 *    lamp-sten/src/main/scala/lamp/device.scala:88-111) and must lamp_tensor_release().
 *  - `_out` variants write into their first argument, a trailing `_` means in place.
 *  - scalar types are ATen's scalar-type bytes (lamp-sten/.../STen.scala:726-731):
 *    0 u8, 1 i8, 2 i16, 3 i32, 4 i64, 5 f16, 6 f32, 7 f64, 11 bool, 15 bf16.
 *    Compute kernels exist for f32, f64, bf16 and f16 (+ i64/i32/u8/bool where indices or
 *    masks are involved).
 *  - device type 0 = host, 1 = GPU (lamp's "cuda", STen.scala:757-759).  This is a GPU
 *    backend: there is no CPU FALLBACK - a GPU tensor never computes on the host and nothing
 *    runs when no MI355X is visible.  Tensors that LIVE in host memory (lamp's `CPU` device,
 *    device.scala:138: arrays in and out, pinned staging, mmap'ed files, scalars, index lists)
 *    support what lamp does with them around the hot path, where they live: construction,
 *    views, copy / cast, element-wise arithmetic and functions, comparisons, sum / mean /
 *    norm / max / min, mm (f32 / f64), the gradient-bucket pack / unpack.  Every other entry
 *    point (convolution, norms, softmax, losses, attention, optimisers, index ops ...) needs
 *    device tensors and says so.
 *  - kernels launch on the calling thread's current stream of the current device
 *    (lamp-sten/.../device.scala:119-129, 199-213).
 */
#ifndef LAMP_HIP_H
#define LAMP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LAMP_MAX_DIMS 8

typedef struct lamp_tensor lamp_tensor;
typedef struct lamp_stream lamp_stream;
typedef struct lamp_comm lamp_comm;
typedef struct lamp_graph lamp_graph;

#define LAMP_U8 0
#define LAMP_I8 1
#define LAMP_I16 2
#define LAMP_I32 3
#define LAMP_I64 4
#define LAMP_F16 5
#define LAMP_F32 6
#define LAMP_F64 7
#define LAMP_BOOL 11
#define LAMP_BF16 15

#define LAMP_DEVICE_CPU (-1)

/* ------------------------------------------------------------------------------------------
 * errors / library
 * ------------------------------------------------------------------------------------------ */
const char* lamp_last_error(void);
const char* lamp_version(void);

/* ------------------------------------------------------------------------------------------
 * devices and streams   (aten.Tensor.hasCuda/getNumGPUs, aten.CudaStream.*:
 * lamp-sten/src/main/scala/lamp/device.scala:178-217, STen.scala:636-639)
 * ------------------------------------------------------------------------------------------ */
int lamp_has_gpu(int* out);
int lamp_get_num_gpus(int* out);
int lamp_get_device(int* out);                 /* cudaGetDevice */
int lamp_set_device(int device);               /* cudaSetDevice */
int lamp_device_synchronize(void);
/* Work the library may defer so that it can be batched (today: the reductions of the bf16 convolutions' weight-gradient partial sums,
 * one launch for all layers of a backward pass) runs at the latest when a pointer into the affected tensor is requested - results
 * never depend on it.  This call runs it NOW on the streams it was registered on: the host autograd calls it at the end of
 * backprop (Variable.backprop, autograd.scala:212-260) so that the whole batch lands in one launch; lamp_device_synchronize and
 * lamp_stream_synchronize include it. */
int lamp_flush_deferred(void);
int lamp_device_name(char* buf, int buflen);
int lamp_device_num_cus(int* out);
int lamp_stream_get_current(int device, lamp_stream** out);      /* getCurrentCUDAStream */
int lamp_stream_get_default(int device, lamp_stream** out);      /* getDefaultCUDAStream */
int lamp_stream_get_from_pool(int high_priority, int device, lamp_stream** out); /* getStreamFromPool */
int lamp_stream_set_current(lamp_stream* s);                     /* setCurrentCUDAStream */
int lamp_stream_synchronize(lamp_stream* s);
int lamp_stream_wait_stream(lamp_stream* waiter, lamp_stream* on); /* event record + wait */
int lamp_stream_release(lamp_stream* s);
/* the tensor's storage is also used by work on `s` (NULL = the caller's current stream): the caching allocator will not hand the
 * block out again before that work is done.  Needed whenever a tensor allocated under one current stream is consumed under another
 * (Device.withOtherStream, device.scala:199-213). */
int lamp_tensor_record_stream(const lamp_tensor* t, lamp_stream* s_or_null);
int lamp_stream_native(lamp_stream* s, void** hip_stream_out);
/* RNG (aten.Tensor.manual_seed*, device.scala:149,170,215; umap.scala:180) */
int lamp_manual_seed(uint64_t seed);
/* aten.Tensor.allowtf32: accepted for source compatibility; gfx950 has no reduced precision
 * f32 MFMA so f32 GEMMs are always exact f32 (example-autoregressivelm/.../main.scala:18) */
int lamp_allow_tf32(int flag);

/* per-kernel-class HIP-event timing for the roofline line of bench.py: when enabled every tagged
 * launch is bracketed by two events on its own stream; the report has one line per tag:
 * "tag launches total_ms algorithmic_flops_per_launch algorithmic_bytes_per_launch" */
int lamp_kernel_timer_enable(int on);
int lamp_kernel_timer_report(char* buf, int buflen);
/* time only launches with this tag (NULL or "" = all); set while the timers are disabled */
int lamp_kernel_timer_filter(const char* tag);
/* median elapsed time of an (event, empty kernel, event) bracket on the current stream, microseconds */
int lamp_kernel_timer_calibrate(double* out_us);

/* "Kernels the library does not schedule are (delta = +1) / are no longer (-1) running on this device": while the count is positive,
 * kernels whose workgroups wait for each other (the one-pass batch-norm backward) take their non-waiting two-kernel form.  The
 * data-parallel step brackets its overlapped RCCL all-reduce with it (distributed/package.scala:690-719); a caller that runs its own
 * side-stream work concurrently with backward does the same. */
int lamp_device_shared_hint(int device, int delta);
/* Batch-norm backward form: -1 the default rule (LAMP_BN_FUSED_BWD, sizes, sharing), 0 always the two kernels, 1 one pass where the
 * geometry qualifies, 2 the same even while the device is marked shared (tests: proves the waiting kernel under CU pressure). */
int lamp_bn_backward_mode(int mode);
/* Test tool: `workgroups` single-wave workgroups that each spin for `microseconds` on `s` (NULL: the current stream) - with the
 * register-file-filling kernels of the training step that is `workgroups` compute units taken away for that long. */
int lamp_debug_occupy_cus(int workgroups, double microseconds, lamp_stream* s_or_null);

/* HIP graph capture of the calling thread's current stream (launch-bound training steps) */
int lamp_graph_begin_capture(void);
int lamp_graph_end_capture(lamp_graph** out);
int lamp_graph_launch(lamp_graph* g);
int lamp_graph_is_capturing(int* out);   /* 1 between lamp_graph_begin_capture and _end_capture on this thread: work issued now is recorded, not run */
int lamp_graph_release(lamp_graph* g);

/* ------------------------------------------------------------------------------------------
 * allocation registry  (aten.TensorTrace / TensorOptionsTrace:
 * lamp-sten/src/main/scala/lamp/TensorLogger.scala:13-62,200-201; leak check
 * lamp-data/src/test/scala/lamp/data/mlp.test.scala:180-188)
 * ------------------------------------------------------------------------------------------ */
int lamp_live_tensor_count(int64_t* out);
/* aten.TensorTrace (TensorLogger.scala:13-62): records of the handles created while the trace is on */
#define LAMP_TRACE_RECORD 14   /* int64 per record: id, birth ns, scalar type, device (-1 host), ndim, sizes[8], bytes */
int lamp_tensor_trace_enable(int on);
int lamp_tensor_trace_list(int64_t* records_or_null, int64_t capacity, int64_t* count);   /* count = live traced handles (may exceed capacity) */
/* Tensor.from_file / tensors_from_file (STen.scala:115-194): read-only mmap of [offset, offset + length) of a file (offset a multiple of
 * 4096, mlock when pin) cut into n one-dimensional HOST tensors (types[i], byte offsets[i] aligned to 8, byte lengths[i]) that share the
 * mapping; it is unmapped with the last handle. */
int lamp_tensors_from_file(lamp_tensor** outs, const char* path, int64_t offset, int64_t length, int pin, const int64_t* types,
                           const int64_t* offsets, const int64_t* lengths, int n);
int lamp_allocator_stats(int device, int64_t* reserved_bytes, int64_t* in_use_bytes, int64_t* n_device_mallocs);
int lamp_allocator_trim(int device);
/* how many frees had to wait for another stream (record_stream'ed blocks) since start-up */
int lamp_allocator_deferred_frees(int device, int64_t* out);

/* ------------------------------------------------------------------------------------------
 * tensor handles, metadata, host <-> device   (aten.Tensor instance methods used by
 * STen.scala:845-1000 and TensorHelpers.scala; factories STen.scala:215-330)
 * ------------------------------------------------------------------------------------------ */
int lamp_tensor_release(lamp_tensor* t);
int lamp_tensor_release_all(lamp_tensor** ts, int n);                    /* Tensor.releaseAll */
int lamp_tensor_retain(const lamp_tensor* t, lamp_tensor** out);
int lamp_tensor_ndim(const lamp_tensor* t, int* out);
int lamp_tensor_sizes(const lamp_tensor* t, int64_t* out /* LAMP_MAX_DIMS */);
int lamp_tensor_strides(const lamp_tensor* t, int64_t* out);
int lamp_tensor_numel(const lamp_tensor* t, int64_t* out);
int lamp_tensor_element_size(const lamp_tensor* t, int64_t* out);
int lamp_tensor_scalar_type(const lamp_tensor* t, int* out);            /* scalarTypeByte */
int lamp_tensor_device(const lamp_tensor* t, int* out);                 /* -1 host, else GPU index */
int lamp_tensor_is_contiguous(const lamp_tensor* t, int* out);
int lamp_tensor_is_pinned(const lamp_tensor* t, int* out);
int lamp_tensor_data_ptr(const lamp_tensor* t, void** out);
int lamp_tensor_storage_id(const lamp_tensor* t, uint64_t* out);        /* views share it */

int lamp_empty(lamp_tensor** out, const int64_t* sizes, int ndim, int dtype, int device);
int lamp_zeros(lamp_tensor** out, const int64_t* sizes, int ndim, int dtype, int device);
int lamp_ones(lamp_tensor** out, const int64_t* sizes, int ndim, int dtype, int device);
int lamp_full(lamp_tensor** out, const int64_t* sizes, int ndim, double value, int dtype, int device);
int lamp_zeros_like(lamp_tensor** out, const lamp_tensor* t);
int lamp_ones_like(lamp_tensor** out, const lamp_tensor* t);
int lamp_scalar_tensor(lamp_tensor** out, double value, int dtype, int device);   /* Tensor.scalarDouble */
int lamp_scalar_tensor_l(lamp_tensor** out, int64_t value, int dtype, int device); /* Tensor.scalarLong */
int lamp_arange(lamp_tensor** out, double start, double end, double step, int dtype, int device);
int lamp_eye(lamp_tensor** out, int64_t n, int64_t m, int dtype, int device);
/* wrap caller-owned memory without copying (from_file / pinned staging / interop) */
int lamp_from_blob(lamp_tensor** out, void* data, const int64_t* sizes, const int64_t* strides, int ndim,
                   int dtype, int device);
/* copyFrom<Type>Array / copyTo<Type>Array (TensorHelpers.scala:44-345): raw bytes of the
 * tensor's own dtype, row-major.  For GPU tensors these stage through the current stream
 * and synchronise it. */
int lamp_copy_from_host(lamp_tensor* dst, const void* src, size_t nbytes);
int lamp_copy_to_host(const lamp_tensor* src, void* dst, size_t nbytes);
int lamp_clone(lamp_tensor** out, const lamp_tensor* t);
int lamp_contiguous(lamp_tensor** out, const lamp_tensor* t);
int lamp_copy_(lamp_tensor* dst, const lamp_tensor* src, int non_blocking);  /* copyFrom, STen.scala:931-944 */
/* value.to(options, non_blocking, copy) (device.scala:221-225; STen.scala:1880) */
int lamp_to(lamp_tensor** out, const lamp_tensor* t, int dtype, int device, int non_blocking, int copy);
int lamp_cast(lamp_tensor** out, const lamp_tensor* t, int dtype);      /* _cast_Float/Double/Long/... */
int lamp_pin_memory(lamp_tensor** out, const lamp_tensor* t);
int lamp_item(const lamp_tensor* t, double* out);                        /* numel==1, syncs */

/* ------------------------------------------------------------------------------------------
 * views   (ATen.t/transpose/select/slice/narrow/_unsafe_view/reshape/flatten/expand_as/
 * squeeze/unsqueeze/cat/stack: STen.scala:210-213,956-971,1374-1380,1472-1491,1740,1766-1775;
 * ops.scala:15-118)
 * ------------------------------------------------------------------------------------------ */
int lamp_view(lamp_tensor** out, const lamp_tensor* t, const int64_t* sizes, int ndim);   /* view/_unsafe_view, one -1 allowed */
int lamp_reshape(lamp_tensor** out, const lamp_tensor* t, const int64_t* sizes, int ndim);
int lamp_flatten(lamp_tensor** out, const lamp_tensor* t, int64_t start_dim, int64_t end_dim);
int lamp_transpose(lamp_tensor** out, const lamp_tensor* t, int64_t dim0, int64_t dim1);
int lamp_t(lamp_tensor** out, const lamp_tensor* t);
int lamp_select(lamp_tensor** out, const lamp_tensor* t, int64_t dim, int64_t index);
int lamp_slice(lamp_tensor** out, const lamp_tensor* t, int64_t dim, int64_t start, int64_t end, int64_t step);
int lamp_narrow(lamp_tensor** out, const lamp_tensor* t, int64_t dim, int64_t start, int64_t length);
/* ATen.as_strided: a view of t's storage with this geometry; storage_offset in elements from the start of the storage */
int lamp_as_strided(lamp_tensor** out, const lamp_tensor* t, const int64_t* sizes, const int64_t* strides, int ndim, int64_t storage_offset);
int lamp_expand(lamp_tensor** out, const lamp_tensor* t, const int64_t* sizes, int ndim);
int lamp_expand_as(lamp_tensor** out, const lamp_tensor* t, const lamp_tensor* other);
int lamp_squeeze(lamp_tensor** out, const lamp_tensor* t, int64_t dim);   /* dim = INT64_MIN: all */
int lamp_unsqueeze(lamp_tensor** out, const lamp_tensor* t, int64_t dim);
int lamp_cat(lamp_tensor** out, lamp_tensor* const* ts, int n, int64_t dim);
int lamp_cat_out(lamp_tensor* out, lamp_tensor* const* ts, int n, int64_t dim);
/* outs[i] = x.chunk(n, dim)[i].contiguous() for equal chunks: what `ATen.chunk` / `slice` followed by `contiguous` give (STen.scala: slice, chunk), in one launch */
int lamp_chunk_contiguous(lamp_tensor** outs, const lamp_tensor* x, int n, int64_t dim);
int lamp_stack(lamp_tensor** out, lamp_tensor* const* ts, int n, int64_t dim);

/* ------------------------------------------------------------------------------------------
 * fills   (fill__0, zero_: STen.scala:1382,1406)
 * ------------------------------------------------------------------------------------------ */
int lamp_fill_(lamp_tensor* t, double value);
int lamp_zero_(lamp_tensor* t);

/* ------------------------------------------------------------------------------------------
 * element-wise, broadcasting   (ATen.add_0/sub_0/mul_0/div_0 + scalar + _out + in-place,
 * addcmul_out/addcdiv_out: STen.scala:365-468,1110-1217,1242-1264; ops.scala:511-621)
 *   out = a + alpha*b  etc.
 * ------------------------------------------------------------------------------------------ */
int lamp_add(lamp_tensor** out, const lamp_tensor* a, const lamp_tensor* b, double alpha);
int lamp_sub(lamp_tensor** out, const lamp_tensor* a, const lamp_tensor* b, double alpha);
int lamp_mul(lamp_tensor** out, const lamp_tensor* a, const lamp_tensor* b);
int lamp_div(lamp_tensor** out, const lamp_tensor* a, const lamp_tensor* b);
int lamp_add_scalar(lamp_tensor** out, const lamp_tensor* a, double b, double alpha);
int lamp_sub_scalar(lamp_tensor** out, const lamp_tensor* a, double b, double alpha);
int lamp_mul_scalar(lamp_tensor** out, const lamp_tensor* a, double b);
int lamp_div_scalar(lamp_tensor** out, const lamp_tensor* a, double b);
int lamp_add_out(lamp_tensor* out, const lamp_tensor* a, const lamp_tensor* b, double alpha);
int lamp_sub_out(lamp_tensor* out, const lamp_tensor* a, const lamp_tensor* b, double alpha);
int lamp_mul_out(lamp_tensor* out, const lamp_tensor* a, const lamp_tensor* b);
int lamp_div_out(lamp_tensor* out, const lamp_tensor* a, const lamp_tensor* b);
int lamp_add_(lamp_tensor* self, const lamp_tensor* b, double alpha);    /* += */
/* acc += scale * x for acc and x of different floating types (sum taken in f64): lamp's loss accumulators are f64 scalars
 * whatever the model type (IOLoops.scala:715 `STen.scalarDouble(0, options)`, SupervisedModel.scala:207) */
int lamp_add_scaled_mixed_(lamp_tensor* acc, const lamp_tensor* x, double scale);
int lamp_sub_(lamp_tensor* self, const lamp_tensor* b, double alpha);    /* -= */
int lamp_mul_(lamp_tensor* self, const lamp_tensor* b);                  /* *= */
int lamp_div_(lamp_tensor* self, const lamp_tensor* b);                  /* /= */
int lamp_add_scalar_(lamp_tensor* self, double b, double alpha);
int lamp_mul_scalar_(lamp_tensor* self, double b);
int lamp_addcmul_out(lamp_tensor* out, const lamp_tensor* self, const lamp_tensor* t1, const lamp_tensor* t2, double value);
/* a * b + c in one pass, the product rounded to the tensor type before the addition: the values of lamp's chain `(a * b) + c`
 * (Transformer.scala:244-247: attention * scale + input) */
int lamp_mul_add(lamp_tensor** out, const lamp_tensor* a, const lamp_tensor* b, const lamp_tensor* c);
int lamp_addcdiv_out(lamp_tensor* out, const lamp_tensor* self, const lamp_tensor* t1, const lamp_tensor* t2, double value);
int lamp_maximum(lamp_tensor** out, const lamp_tensor* a, const lamp_tensor* b);   /* max_2, knn/package.scala:28 */
int lamp_minimum(lamp_tensor** out, const lamp_tensor* a, const lamp_tensor* b);
int lamp_pow_scalar(lamp_tensor** out, const lamp_tensor* a, double exponent);     /* pow_2 */
int lamp_pow_tensor(lamp_tensor** out, const lamp_tensor* a, const lamp_tensor* exponent);
int lamp_where(lamp_tensor** out, const lamp_tensor* cond, const lamp_tensor* a, const lamp_tensor* b); /* where_0 */
int lamp_masked_fill(lamp_tensor** out, const lamp_tensor* a, const lamp_tensor* mask, double value);
/* comparisons -> bool tensors (lt_0/le_0/ne_1/eq/gt/ge: STen.scala:1013-1030,1610-1665) */
int lamp_lt_scalar(lamp_tensor** out, const lamp_tensor* a, double b);
int lamp_le_scalar(lamp_tensor** out, const lamp_tensor* a, double b);
int lamp_gt_scalar(lamp_tensor** out, const lamp_tensor* a, double b);
int lamp_ge_scalar(lamp_tensor** out, const lamp_tensor* a, double b);
int lamp_eq_scalar(lamp_tensor** out, const lamp_tensor* a, double b);
int lamp_ne_scalar(lamp_tensor** out, const lamp_tensor* a, double b);
int lamp_lt(lamp_tensor** out, const lamp_tensor* a, const lamp_tensor* b);
int lamp_le(lamp_tensor** out, const lamp_tensor* a, const lamp_tensor* b);
int lamp_gt(lamp_tensor** out, const lamp_tensor* a, const lamp_tensor* b);
int lamp_ge(lamp_tensor** out, const lamp_tensor* a, const lamp_tensor* b);
int lamp_eq(lamp_tensor** out, const lamp_tensor* a, const lamp_tensor* b);
int lamp_ne(lamp_tensor** out, const lamp_tensor* a, const lamp_tensor* b);
int lamp_logical_not(lamp_tensor** out, const lamp_tensor* a);

/* unary (STen.scala:1266-1320; activations ops.scala:854-1032).  In-place twins take `_`. */
int lamp_relu(lamp_tensor** out, const lamp_tensor* a);
int lamp_relu_(lamp_tensor* a);
int lamp_leaky_relu(lamp_tensor** out, const lamp_tensor* a, double negative_slope);
int lamp_gelu(lamp_tensor** out, const lamp_tensor* a);                   /* exact erf form */
int lamp_gelu_backward(lamp_tensor** out, const lamp_tensor* grad, const lamp_tensor* self);
int lamp_sigmoid(lamp_tensor** out, const lamp_tensor* a);
int lamp_sigmoid_backward(lamp_tensor** out, const lamp_tensor* grad, const lamp_tensor* output);
int lamp_tanh(lamp_tensor** out, const lamp_tensor* a);
int lamp_tanh_backward(lamp_tensor** out, const lamp_tensor* grad, const lamp_tensor* output);
int lamp_hardswish(lamp_tensor** out, const lamp_tensor* a);
int lamp_hardswish_backward(lamp_tensor** out, const lamp_tensor* grad, const lamp_tensor* self);
int lamp_softplus(lamp_tensor** out, const lamp_tensor* a, double beta, double threshold);
int lamp_softplus_backward(lamp_tensor** out, const lamp_tensor* grad, const lamp_tensor* self, double beta, double threshold);
int lamp_exp(lamp_tensor** out, const lamp_tensor* a);
int lamp_exp_(lamp_tensor* a);
int lamp_log(lamp_tensor** out, const lamp_tensor* a);
int lamp_log1p(lamp_tensor** out, const lamp_tensor* a);
int lamp_sqrt(lamp_tensor** out, const lamp_tensor* a);
int lamp_sqrt_(lamp_tensor* a);
int lamp_square(lamp_tensor** out, const lamp_tensor* a);
int lamp_reciprocal(lamp_tensor** out, const lamp_tensor* a);
int lamp_reciprocal_(lamp_tensor* a);
int lamp_neg(lamp_tensor** out, const lamp_tensor* a);
int lamp_abs(lamp_tensor** out, const lamp_tensor* a);
int lamp_sign(lamp_tensor** out, const lamp_tensor* a);
int lamp_sin(lamp_tensor** out, const lamp_tensor* a);
int lamp_cos(lamp_tensor** out, const lamp_tensor* a);
int lamp_tan(lamp_tensor** out, const lamp_tensor* a);
int lamp_atan(lamp_tensor** out, const lamp_tensor* a);
/* the remaining element-wise members of the aten surface lamp-sten binds (STen.scala:1200-1700): math, in-place forms, logicals */
int lamp_acos(lamp_tensor** out, const lamp_tensor* a);
int lamp_asin(lamp_tensor** out, const lamp_tensor* a);
int lamp_ceil(lamp_tensor** out, const lamp_tensor* a);
int lamp_floor(lamp_tensor** out, const lamp_tensor* a);
int lamp_round(lamp_tensor** out, const lamp_tensor* a);      /* half to even, as ATen */
int lamp_expm1(lamp_tensor** out, const lamp_tensor* a);
int lamp_log10(lamp_tensor** out, const lamp_tensor* a);
int lamp_atan2(lamp_tensor** out, const lamp_tensor* a, const lamp_tensor* b);
int lamp_remainder(lamp_tensor** out, const lamp_tensor* a, const lamp_tensor* b);   /* sign of the divisor */
int lamp_remainder_scalar(lamp_tensor** out, const lamp_tensor* a, double b);
int lamp_nan_to_num(lamp_tensor** out, const lamp_tensor* a, double nan);            /* infinities -> largest finite values */
int lamp_abs_(lamp_tensor* a);
int lamp_acos_(lamp_tensor* a);
int lamp_asin_(lamp_tensor* a);
int lamp_atan_(lamp_tensor* a);
int lamp_ceil_(lamp_tensor* a);
int lamp_floor_(lamp_tensor* a);
int lamp_cos_(lamp_tensor* a);
int lamp_sin_(lamp_tensor* a);
int lamp_tan_(lamp_tensor* a);
int lamp_tanh_(lamp_tensor* a);
int lamp_sigmoid_(lamp_tensor* a);
int lamp_log_(lamp_tensor* a);
int lamp_log1p_(lamp_tensor* a);
int lamp_square_(lamp_tensor* a);
int lamp_leaky_relu_(lamp_tensor* a, double slope);
int lamp_logical_and(lamp_tensor** out, const lamp_tensor* a, const lamp_tensor* b);
int lamp_logical_or(lamp_tensor** out, const lamp_tensor* a, const lamp_tensor* b);
int lamp_logical_xor(lamp_tensor** out, const lamp_tensor* a, const lamp_tensor* b);
int lamp_isnan(lamp_tensor** out, const lamp_tensor* a);
int lamp_isfinite(lamp_tensor** out, const lamp_tensor* a);
/* fused forms of lamp's op backward closures (same arithmetic as the ATen call chains they
 * replace; see DESIGN.md "fused backward closures"):
 *   relu:  out += p * (x < 0 ? 0 : 1)      ops.scala:918-935 (gradient at x == 0 is 1)
 *   leaky: out += p * (x < 0 ? slope : 1)  ops.scala:936-953 */
int lamp_relu_backward_accumulate_(lamp_tensor* out, const lamp_tensor* p, const lamp_tensor* x, double negative_slope);
/* the same product written to a fresh tensor (first accumulation into a still-zero gradient: 0 + v == v) */
int lamp_relu_backward(lamp_tensor** out, const lamp_tensor* p, const lamp_tensor* x, double negative_slope);

/* ------------------------------------------------------------------------------------------
 * reductions   (sum_0/sum_1/mean_1/norm_3/var_mean/argmax/max/min: STen.scala:1336-1352,
 * 1493-1501,1524-1540,1565-1585,987; TensorHelpers.unbroadcast TensorHelpers.scala:7-41)
 * ------------------------------------------------------------------------------------------ */
int lamp_sum_all(lamp_tensor** out, const lamp_tensor* a);
int lamp_sum_dims(lamp_tensor** out, const lamp_tensor* a, const int64_t* dims, int ndims, int keepdim);
int lamp_mean_all(lamp_tensor** out, const lamp_tensor* a);
int lamp_mean_dims(lamp_tensor** out, const lamp_tensor* a, const int64_t* dims, int ndims, int keepdim);
int lamp_norm2_dims(lamp_tensor** out, const lamp_tensor* a, const int64_t* dims, int ndims, int keepdim);
int lamp_var_mean_dims(lamp_tensor** var_out, lamp_tensor** mean_out, const lamp_tensor* a, const int64_t* dims,
                       int ndims, int unbiased, int keepdim);
int lamp_max_all(lamp_tensor** out, const lamp_tensor* a);
int lamp_min_all(lamp_tensor** out, const lamp_tensor* a);
int lamp_argmax(lamp_tensor** out, const lamp_tensor* a, int64_t dim, int keepdim);
int lamp_unbroadcast(lamp_tensor** out, const lamp_tensor* p, const int64_t* target_sizes, int ndim);

/* ------------------------------------------------------------------------------------------
 * GEMM family   (ATen.mm/bmm/addmm/baddbmm/matmul + aten-scala's custom natives
 * Tensor.addmm_out_transposed1/2 and Tensor.baddbmm_out_transposed1/2:
 * STen.scala:391-449,1146,1220-1240; ops.scala:665-724)
 *   addmm_out_transposed1: out = beta*self + alpha * (a^T . b)
 *   addmm_out_transposed2: out = beta*self + alpha * (a . b^T)
 * bf16 -> MFMA 16x16x32 bf16, fp32 accumulate; f32 -> exact f32 MFMA; f64 -> f64 MFMA.
 * ------------------------------------------------------------------------------------------ */
int lamp_mm(lamp_tensor** out, const lamp_tensor* a, const lamp_tensor* b);
int lamp_mm_out(lamp_tensor* out, const lamp_tensor* a, const lamp_tensor* b);
int lamp_bmm(lamp_tensor** out, const lamp_tensor* a, const lamp_tensor* b);
int lamp_bmm_out(lamp_tensor* out, const lamp_tensor* a, const lamp_tensor* b);
int lamp_matmul(lamp_tensor** out, const lamp_tensor* a, const lamp_tensor* b);
int lamp_addmm(lamp_tensor** out, const lamp_tensor* self, const lamp_tensor* a, const lamp_tensor* b, double beta, double alpha);
int lamp_addmm_out(lamp_tensor* out, const lamp_tensor* self, const lamp_tensor* a, const lamp_tensor* b, double beta, double alpha);
int lamp_baddbmm(lamp_tensor** out, const lamp_tensor* self, const lamp_tensor* a, const lamp_tensor* b, double beta, double alpha);
int lamp_addmm_out_transposed1(lamp_tensor* out, const lamp_tensor* self, const lamp_tensor* a, const lamp_tensor* b, double beta, double alpha);
int lamp_addmm_out_transposed2(lamp_tensor* out, const lamp_tensor* self, const lamp_tensor* a, const lamp_tensor* b, double beta, double alpha);
int lamp_baddbmm_out_transposed1(lamp_tensor* out, const lamp_tensor* self, const lamp_tensor* a, const lamp_tensor* b, double beta, double alpha);
int lamp_baddbmm_out_transposed2(lamp_tensor* out, const lamp_tensor* self, const lamp_tensor* a, const lamp_tensor* b, double beta, double alpha);
/* fused Linear forward: y = x.W + bias[1,out] (lamp-core/.../nn/Linear.scala:19-33 issues
 * mm then a broadcast add; this is the same arithmetic with the add in the GEMM epilogue) */
int lamp_linear_bias(lamp_tensor** out, const lamp_tensor* x, const lamp_tensor* w, const lamp_tensor* bias_or_null);

/* ------------------------------------------------------------------------------------------
 * convolution / pooling   (ATen.convolution, convolution_backward(output_mask[3]),
 * avg_pool2d(+_backward), max_pool2d_with_indices(+_backward): ops.scala:1547-1651,
 * 1721-1825).  NCHW / NCL contiguous, groups supported, transposed supported.
 * ------------------------------------------------------------------------------------------ */
int lamp_convolution(lamp_tensor** out, const lamp_tensor* x, const lamp_tensor* w, const lamp_tensor* bias_or_null,
                     const int64_t* stride, const int64_t* padding, const int64_t* dilation, int nspatial,
                     int transposed, const int64_t* output_padding, int64_t groups);
/* out3 = {grad_input, grad_weight, grad_bias}; entries whose mask is 0 come back NULL */
int lamp_convolution_backward(lamp_tensor* out3[3], const lamp_tensor* grad_out, const lamp_tensor* x,
                              const lamp_tensor* w, const int64_t* stride, const int64_t* padding,
                              const int64_t* dilation, int nspatial, int transposed,
                              const int64_t* output_padding, int64_t groups, const uint8_t mask[3]);
/* grad_input of the (non-transposed) convolution + addend, values of lamp_convolution_backward followed by lamp_add (each rounded to
 * the tensor's dtype), in one pass where the kernel has the epilogue: autograd.scala:66-84 accumulates the partial derivatives of a
 * Variable with several consumers (a residual block's input) by `+=`; this is that `+=` folded into the second contribution. */
int lamp_convolution_backward_input_add(lamp_tensor** out, const lamp_tensor* grad_out, const lamp_tensor* x,
                                        const lamp_tensor* w, const int64_t* stride, const int64_t* padding,
                                        const int64_t* dilation, int nspatial, const int64_t* output_padding,
                                        int64_t groups, const lamp_tensor* addend);
/* grad_input of TWO (non-transposed) convolutions of ONE input x, summed (+ addend when given): what autograd.scala:66-84 accumulates into
 * the input of lamp's residual block (cnn.scala:16-20: a 3x3 branch and the 1x1 shortcut) from its two consumers.  Where one kernel takes
 * both gradients (bf16, the narrow layers: 3x3 pad 1 + 1x1 pad 0 of equal stride) the two products are summed in f32 and rounded ONCE;
 * everywhere else the value is lamp_convolution_backward (b) followed by lamp_convolution_backward_input_add (a), each rounded to the dtype. */
int lamp_convolution_backward_input_pair(lamp_tensor** out, const lamp_tensor* x, const lamp_tensor* grad_out_a, const lamp_tensor* w_a,
                                         const int64_t* stride_a, const int64_t* padding_a, const int64_t* dilation_a,
                                         const lamp_tensor* grad_out_b, const lamp_tensor* w_b, const int64_t* stride_b,
                                         const int64_t* padding_b, const int64_t* dilation_b, int nspatial, int64_t groups,
                                         const lamp_tensor* addend_or_null);
/* grad_weight of the same two convolutions: out2 = {dW_a, dW_b}, the values of two lamp_convolution_backward calls with mask (0, 1, 0); one
 * launch that stages x once where a kernel takes both output gradients (bf16 narrow layers, 3x3 pad 1 + 1x1 pad 0, at most 16 output
 * channels together).  w_a / w_b give the filters' shapes and dtype. */
int lamp_convolution_backward_weight_pair(lamp_tensor* out2[2], const lamp_tensor* x, const lamp_tensor* grad_out_a, const lamp_tensor* w_a,
                                          const int64_t* stride_a, const int64_t* padding_a, const int64_t* dilation_a,
                                          const lamp_tensor* grad_out_b, const lamp_tensor* w_b, const int64_t* stride_b,
                                          const int64_t* padding_b, const int64_t* dilation_b, int nspatial, int64_t groups);
/* TWO (non-transposed) convolutions of ONE input: out2 = {convolution(x, w_a, bias_a, geometry a), convolution(x, w_b, bias_b, geometry b)} - the
 * two branches of lamp's residual block both start with a Conv2D on the block's input (example-cifar100 cnn.scala:16-20, 38-78: 3x3 and the
 * 1x1 shortcut).  Values (and the batch-norm statistics hand-off of each output) are those of two lamp_convolution calls; where a kernel
 * holds the staged input for both products (bf16 3x3 + 1x1 on 8x8 maps with equal output channels) they are one launch. */
int lamp_convolution_pair(lamp_tensor* out2[2], const lamp_tensor* x, const lamp_tensor* w_a, const lamp_tensor* bias_a_or_null,
                          const int64_t* stride_a, const int64_t* padding_a, const int64_t* dilation_a, const lamp_tensor* w_b,
                          const lamp_tensor* bias_b_or_null, const int64_t* stride_b, const int64_t* padding_b,
                          const int64_t* dilation_b, int nspatial, int64_t groups);
int lamp_avg_pool2d(lamp_tensor** out, const lamp_tensor* x, int64_t kernel, int64_t stride, int64_t padding,
                    int ceil_mode, int count_include_pad);
int lamp_avg_pool2d_backward(lamp_tensor** out, const lamp_tensor* grad_out, const lamp_tensor* x, int64_t kernel,
                             int64_t stride, int64_t padding, int ceil_mode, int count_include_pad);
int lamp_max_pool2d_with_indices(lamp_tensor** out, lamp_tensor** indices, const lamp_tensor* x, int64_t kernel,
                                 int64_t stride, int64_t padding, int64_t dilation, int ceil_mode);
int lamp_max_pool2d_with_indices_backward(lamp_tensor** out, const lamp_tensor* grad_out, const lamp_tensor* x,
                                          int64_t kernel, int64_t stride, int64_t padding, int64_t dilation,
                                          int ceil_mode, const lamp_tensor* indices);
/* max_pool1d_with_indices over [N, C, L] (MaxPool1D, ops.scala:1658-1715); indices are positions along L */
int lamp_max_pool1d_with_indices(lamp_tensor** out, lamp_tensor** indices, const lamp_tensor* x, int64_t kernel,
                                 int64_t stride, int64_t padding, int64_t dilation, int ceil_mode);
int lamp_max_pool1d_with_indices_backward(lamp_tensor** out, const lamp_tensor* grad_out, const lamp_tensor* x,
                                          int64_t kernel, int64_t stride, int64_t padding, int64_t dilation,
                                          int ceil_mode, const lamp_tensor* indices);

/* ------------------------------------------------------------------------------------------
 * normalisation   (ATen.native_batch_norm(+_backward), native_layer_norm(+_backward):
 * ops.scala:1846-2140).  running_mean/var are updated in place when training
 * (running_var with the unbiased estimate); stats are computed in f32 for bf16 inputs.
 * ------------------------------------------------------------------------------------------ */
int lamp_native_batch_norm(lamp_tensor* out3[3] /* y, save_mean, save_invstd */, const lamp_tensor* x,
                           const lamp_tensor* weight_or_null, const lamp_tensor* bias_or_null,
                           lamp_tensor* running_mean_or_null, lamp_tensor* running_var_or_null, int training,
                           double momentum, double eps);
int lamp_native_batch_norm_backward(lamp_tensor* out3[3] /* dx, dweight, dbias */, const lamp_tensor* grad_out,
                                    const lamp_tensor* x, const lamp_tensor* weight_or_null,
                                    const lamp_tensor* running_mean_or_null, const lamp_tensor* running_var_or_null,
                                    const lamp_tensor* save_mean_or_null, const lamp_tensor* save_invstd_or_null,
                                    int training, double eps, const uint8_t mask[3]);
/* Fused forms of the pair lamp's networks chain everywhere, BatchNorm -> relu (cnn.scala:36-40,
 * MLP.scala:98-120): y = relu(native_batch_norm(x)), rounded to the element type between the two exactly
 * as the unfused pair is, so the results are identical; the backward recomputes the relu mask from x
 * (grad passes where the normalised value is >= 0, ops.scala:918-953) and saves the three elementwise
 * passes of relu / relu-backward.  Not ATen natives: optional extensions in the style of aten-scala's
 * addmm_out_transposed. */
int lamp_native_batch_norm_relu(lamp_tensor* out3[3] /* y, save_mean, save_invstd */, const lamp_tensor* x,
                                const lamp_tensor* weight_or_null, const lamp_tensor* bias_or_null,
                                lamp_tensor* running_mean_or_null, lamp_tensor* running_var_or_null, int training,
                                double momentum, double eps);
int lamp_native_batch_norm_relu_backward(lamp_tensor* out3[3] /* dx, dweight, dbias */, const lamp_tensor* grad_out,
                                         const lamp_tensor* x, const lamp_tensor* weight_or_null, const lamp_tensor* bias_or_null,
                                         const lamp_tensor* running_mean_or_null, const lamp_tensor* running_var_or_null,
                                         const lamp_tensor* save_mean_or_null, const lamp_tensor* save_invstd_or_null,
                                         int training, double eps, const uint8_t mask[3]);
/* relu(native_batch_norm(x) + addend): the tail of lamp's residual block (cnn.scala:11-21,36-46: BatchNorm2D of the right
 * branch, Residual's add, Fun(relu)), each intermediate rounded as the three separate ops round it.  Training mode, maps of
 * >= 64 elements.  The backward returns the four gradients (mask[3] = d addend, which is the relu-masked grad_out). */
int lamp_native_batch_norm_add_relu(lamp_tensor* out3[3] /* y, save_mean, save_invstd */, const lamp_tensor* x,
                                    const lamp_tensor* addend, const lamp_tensor* weight_or_null, const lamp_tensor* bias_or_null,
                                    lamp_tensor* running_mean_or_null, lamp_tensor* running_var_or_null, int training,
                                    double momentum, double eps);
int lamp_native_batch_norm_add_relu_backward(lamp_tensor* out4[4] /* dx, dweight, dbias, daddend */, const lamp_tensor* grad_out,
                                             const lamp_tensor* x, const lamp_tensor* addend, const lamp_tensor* weight_or_null,
                                             const lamp_tensor* bias_or_null, const lamp_tensor* running_mean_or_null,
                                             const lamp_tensor* running_var_or_null, const lamp_tensor* save_mean_or_null,
                                             const lamp_tensor* save_invstd_or_null, int training, double eps, const uint8_t mask[4]);
/* relu(batch_norm(x) + batch_norm2(x2)): the tail of lamp's residual block when BOTH branches end in a batch norm (the right branch's
 * Seq6 and the left branch's Conv2D -> BatchNorm2D, example-cifar100 cnn.scala:36-78, followed by Fun(relu)) as one kernel per direction.
 * Training mode, maps of >= 64 elements.  Forward values are bitwise those of the chain batch_norm2 -> batch_norm_add_relu; out5 =
 * y, save_mean, save_invstd, save_mean2, save_invstd2; out6 = dx, dweight, dbias, dx2, dweight2, dbias2 (mask selects). */
int lamp_native_batch_norm2_add_relu(lamp_tensor* out5[5], const lamp_tensor* x, const lamp_tensor* weight, const lamp_tensor* bias,
                                     lamp_tensor* running_mean, lamp_tensor* running_var, const lamp_tensor* x2, const lamp_tensor* weight2,
                                     const lamp_tensor* bias2, lamp_tensor* running_mean2, lamp_tensor* running_var2, double momentum,
                                     double momentum2, double eps, double eps2);
/* ... followed by AvgPool2D over the whole map -> Flatten -> LogSoftMax: the LAST block of Cnn.resnet and the network's tail (cnn.scala:129-136) in
 * one call.  The block's output has one reader, the pool, and is never written: out5[0] = the log-probabilities [N, C] (bitwise those of
 * lamp_global_avg_pool_log_softmax(lamp_native_batch_norm2_add_relu(...)[0])), the other four as above.  Its backward is
 * lamp_native_batch_norm2_add_relu_backward on the pooled LogSoftMax's input gradient (one value per plane, as an expanded view). */
int lamp_native_batch_norm2_add_relu_pool_log_softmax(lamp_tensor* out5[5], const lamp_tensor* x, const lamp_tensor* weight, const lamp_tensor* bias,
                                                      lamp_tensor* running_mean, lamp_tensor* running_var, const lamp_tensor* x2,
                                                      const lamp_tensor* weight2, const lamp_tensor* bias2, lamp_tensor* running_mean2,
                                                      lamp_tensor* running_var2, double momentum, double momentum2, double eps, double eps2);
int lamp_native_batch_norm2_add_relu_backward(lamp_tensor* out6[6], const lamp_tensor* grad_out, const lamp_tensor* x, const lamp_tensor* weight,
                                              const lamp_tensor* bias, const lamp_tensor* save_mean, const lamp_tensor* save_invstd,
                                              const lamp_tensor* x2, const lamp_tensor* weight2, const lamp_tensor* bias2,
                                              const lamp_tensor* save_mean2, const lamp_tensor* save_invstd2, double eps, double eps2,
                                              const uint8_t mask[6]);
/* The batch norm + relu BETWEEN two convolutions of a residual block (Conv2D -> BatchNorm2D -> relu -> Dropout(0) -> Conv2D, cnn.scala:38-60)
 * applied by the second convolution while it stages its input, so that the normalised tensor is neither written nor re-read:
 *   lamp_batch_norm_affine           the training-mode statistics of x (running statistics updated as lamp_native_batch_norm does) and the
 *                                    f32 table [C, 4] = (mean, invstd * weight, bias, 0) per channel, both rounded as the saved tensors are;
 *   lamp_convolution_bn_relu_input   convolution(relu(batch_norm(x)), w, bias): bitwise the result of the two separate operators;
 *   ..._backward                     (gradient w.r.t. relu(batch_norm(x)), dweight, dbias); the caller continues with
 *                                    lamp_native_batch_norm_relu_backward on out3[0] and x.
 * f32 / f16 / bf16, no transposed convolution.  Geometries outside the kernels that fold the table materialise relu(bn(x)) once. */
int lamp_batch_norm_affine(lamp_tensor* out3[3] /* affine, save_mean, save_invstd */, const lamp_tensor* x, const lamp_tensor* weight,
                           const lamp_tensor* bias, lamp_tensor* running_mean, lamp_tensor* running_var, double momentum, double eps);
int lamp_convolution_bn_relu_input_folds(int* out /* 1: both the forward and the weight-gradient kernel fold the table for this geometry */,
                                         const lamp_tensor* x, const lamp_tensor* w, const int64_t* stride, const int64_t* padding,
                                         const int64_t* dilation, int nspatial, int64_t groups);
int lamp_convolution_bn_relu_input(lamp_tensor** out, const lamp_tensor* x, const lamp_tensor* affine, const lamp_tensor* w, const lamp_tensor* bias_or_null,
                                   const int64_t* stride, const int64_t* padding, const int64_t* dilation, int nspatial, int64_t groups);
int lamp_convolution_bn_relu_input_backward(lamp_tensor* out3[3] /* dactivation, dweight, dbias */, const lamp_tensor* grad_out, const lamp_tensor* x,
                                            const lamp_tensor* affine, const lamp_tensor* w, const int64_t* stride, const int64_t* padding,
                                            const int64_t* dilation, int nspatial, int64_t groups, const uint8_t mask[3]);
int lamp_native_layer_norm(lamp_tensor* out3[3] /* y, mean, rstd */, const lamp_tensor* x,
                           const int64_t* normalized_shape, int nnorm, const lamp_tensor* weight_or_null,
                           const lamp_tensor* bias_or_null, double eps);
int lamp_native_layer_norm_backward(lamp_tensor* out3[3] /* dx, dweight, dbias */, const lamp_tensor* grad_out,
                                    const lamp_tensor* x, const int64_t* normalized_shape, int nnorm,
                                    const lamp_tensor* mean, const lamp_tensor* rstd,
                                    const lamp_tensor* weight_or_null, const lamp_tensor* bias_or_null,
                                    const uint8_t mask[3]);

/* ------------------------------------------------------------------------------------------
 * softmax / losses   (ATen.log_softmax, _log_softmax_backward_data, nll_loss_forward/backward,
 * mse_loss(+_backward): ops.scala:955-975,1176-1304; STen.scala:602-616,1509)
 * reduction: 0 none, 1 mean, 2 sum
 * ------------------------------------------------------------------------------------------ */
int lamp_log_softmax(lamp_tensor** out, const lamp_tensor* x, int64_t dim);
int lamp_log_softmax_backward_data(lamp_tensor** out, const lamp_tensor* grad_out, const lamp_tensor* output, int64_t dim);
/* avg_pool2d(kernel = H = W) -> flatten -> log_softmax(dim 1) of [N, C, H, W] as one operator with the values of the three calls
 * (each stage rounded to the dtype): the tail of Cnn.resnet (cnn.scala:129-136).  One kernel where the planes are whole 16-byte
 * packets, the three calls otherwise.  _backward: grad and output are [N, C], x only gives the shape. */
int lamp_global_avg_pool_log_softmax(lamp_tensor** out, const lamp_tensor* x);
int lamp_global_avg_pool_log_softmax_backward(lamp_tensor** out, const lamp_tensor* grad, const lamp_tensor* output,
                                              const lamp_tensor* x);
/* ... and the NllLoss behind it (SupervisedModel: loss(LogSoftMax(pool(x)), target)): nll_loss_backward followed by the call above, the loss's
 * gradient row built inside the launch (values of the two calls, each stage rounded to the dtype). */
int lamp_global_avg_pool_log_softmax_nll_backward(lamp_tensor** out, const lamp_tensor* grad_loss, const lamp_tensor* target /* i64 [N] */,
                                                  const lamp_tensor* weight_or_null, int64_t reduction, int64_t ignore_index,
                                                  const lamp_tensor* total_weight, const lamp_tensor* output, const lamp_tensor* x);
int lamp_softmax(lamp_tensor** out, const lamp_tensor* x, int64_t dim);
int lamp_nll_loss_forward(lamp_tensor** out, lamp_tensor** total_weight, const lamp_tensor* x,
                          const lamp_tensor* target /* i64 [N] */, const lamp_tensor* weight_or_null,
                          int64_t reduction, int64_t ignore_index);
/* the same, and acc += scale * loss in the same launch (acc: one element of x's dtype; reduction mean or sum): lamp's epoch-loss
 * bookkeeping `acc += loss * numExamples` (IOLoops.scala:714), with the arithmetic of lamp_add_ on the rounded loss */
int lamp_nll_loss_forward_accumulate_(lamp_tensor** out, lamp_tensor** total_weight, const lamp_tensor* x,
                                      const lamp_tensor* target, const lamp_tensor* weight_or_null, int64_t reduction,
                                      int64_t ignore_index, lamp_tensor* acc, double scale);
/* nll_loss_forward (mean or sum; acc_or_null as above) on the log-probabilities x [N, C] that lamp_global_avg_pool_log_softmax made of maps
 * of plane_elems = H * W elements, and from the SAME launch the input gradient of that pooled LogSoftMax for an incoming loss gradient of one -
 * what backprop seeds the loss with (Variable.backprop, autograd.scala:264-282): plane_grad [N, C], where every element of plane (n, c) of
 * lamp_global_avg_pool_log_softmax_nll_backward(ones, ...) equals plane_grad[n][c], bit for bit.  plane_grad is stored class-major (strides
 * [1, N]: its consumer walks one channel at a time).  The caller hands the gradient on as the view expand(plane_grad.unsqueeze(2).unsqueeze(3),
 * [N, C, H, W]) (strides [1, N, 0, 0]); the one-pass batch-norm backward reads such a view without materialising it.
 * *plane_grad is NULL (and the call is the plain forward) for shapes the fused form does not take. */
int lamp_nll_loss_forward_pooled_gradient_(lamp_tensor** out, lamp_tensor** total_weight, lamp_tensor** plane_grad, const lamp_tensor* x,
                                           const lamp_tensor* target /* i64 [N] */, const lamp_tensor* weight_or_null, int64_t reduction,
                                           int64_t ignore_index, lamp_tensor* acc_or_null, double scale, int64_t plane_elems);
int lamp_nll_loss_backward(lamp_tensor** out, const lamp_tensor* grad_out, const lamp_tensor* x,
                           const lamp_tensor* target, const lamp_tensor* weight_or_null, int64_t reduction,
                           int64_t ignore_index, const lamp_tensor* total_weight);
int lamp_mse_loss(lamp_tensor** out, const lamp_tensor* x, const lamp_tensor* target, int64_t reduction);
int lamp_mse_loss_backward(lamp_tensor** out, const lamp_tensor* grad_out, const lamp_tensor* x,
                           const lamp_tensor* target, int64_t reduction);
/* smooth_l1_loss_0 / smooth_l1_loss_backward_0 and binary_cross_entropy_with_logits (SmoothL1Loss, BinaryCrossEntropyWithLogitsLoss:
 * ops.scala:1207-1247, 1309-1367); reduction 0 none / 1 mean / 2 sum; pos_weight may be NULL */
int lamp_smooth_l1_loss(lamp_tensor** out, const lamp_tensor* x, const lamp_tensor* target, int64_t reduction, double beta);
int lamp_smooth_l1_loss_backward(lamp_tensor** out, const lamp_tensor* grad_out, const lamp_tensor* x, const lamp_tensor* target,
                                 int64_t reduction, double beta);
int lamp_binary_cross_entropy_with_logits(lamp_tensor** out, const lamp_tensor* x, const lamp_tensor* target,
                                          const lamp_tensor* pos_weight_or_null, int64_t reduction);

/* ------------------------------------------------------------------------------------------
 * indexing / sampling   (index_select, index_add_0, index, masked_select, repeat_interleave_2,
 * randint, topk, embedding: ops.scala:179-197; BatchStream.scala:548-549; umap.scala:211-227;
 * knn/package.scala:55).  Index tensors are i64 and results are bit-exact.
 * ------------------------------------------------------------------------------------------ */
int lamp_index_select(lamp_tensor** out, const lamp_tensor* a, int64_t dim, const lamp_tensor* index);
/* rows of a PINNED HOST tensor gathered by a device index list straight into a device batch (the GPU reads them over PCIe), converted to
 * out_dtype on the way (-1: the stored type; f32 -> bf16, u8 -> f32 / bf16, f64 -> f32): the minibatch of a host-resident data set
 * (BatchStream.scala:539-556: host gather + pinned staging buffer + copy).  Runs on the current stream of index's device. */
int lamp_index_select_pinned(lamp_tensor** out, const lamp_tensor* pinned, const lamp_tensor* index, int out_dtype);
int lamp_index_add(lamp_tensor** out, const lamp_tensor* self, int64_t dim, const lamp_tensor* index, const lamp_tensor* source);
int lamp_index_add_(lamp_tensor* self, int64_t dim, const lamp_tensor* index, const lamp_tensor* source);
int lamp_masked_select(lamp_tensor** out, const lamp_tensor* a, const lamp_tensor* mask);   /* syncs (output size) */
int lamp_repeat_interleave(lamp_tensor** out, const lamp_tensor* a, int64_t repeats, int64_t dim);
int lamp_repeat_interleave_tensor(lamp_tensor** out, const lamp_tensor* a, const lamp_tensor* repeats /* i64[size(dim)] */, int64_t dim); /* syncs */
/* masked_scatter / gather / scatter_add / index_fill: the ATen calls behind lamp's MaskSelect, ElementWiseMinimum / Maximum, ScatterAdd and
 * IndexFill closures (ops.scala:133-177, 410-434, 2287-2340).  An index outside the tensor raises at the next host wait. */
int lamp_masked_scatter(lamp_tensor** out, const lamp_tensor* self, const lamp_tensor* mask, const lamp_tensor* source);
int lamp_gather(lamp_tensor** out, const lamp_tensor* a, int64_t dim, const lamp_tensor* index);
int lamp_scatter_add(lamp_tensor** out, const lamp_tensor* self, int64_t dim, const lamp_tensor* index, const lamp_tensor* src);
int lamp_diag(lamp_tensor** out, const lamp_tensor* a, int64_t diagonal);          /* ATen.diag: vector -> matrix, matrix -> k-th diagonal (ops.scala:333-350) */
int lamp_cross(lamp_tensor** out, const lamp_tensor* a, const lamp_tensor* b, int64_t dim);   /* ATen.cross along a dimension of size 3 (ops.scala:581-601) */
int lamp_index_fill(lamp_tensor** out, const lamp_tensor* a, int64_t dim, const lamp_tensor* index, double value);
int lamp_topk(lamp_tensor** values, lamp_tensor** indices, const lamp_tensor* a, int64_t k, int64_t dim, int largest, int sorted);
/* Sorting family (kernels/sort.hip; STen.scala:1592 argsort, :1761 sort, :1553 median, :1037-1055 unique, :1034 bincount).  The sort is
 * stable in both directions (equal values keep their order; NaN sorts above every number, as ATen's), which is also a legal result of
 * ATen's unstable sort.  lamp_unique / lamp_bincount have data dependent output sizes: one host synchronisation each. */
int lamp_sort(lamp_tensor** values, lamp_tensor** indices, const lamp_tensor* a, int64_t dim, int descending);
int lamp_argsort(lamp_tensor** out, const lamp_tensor* a, int stable, int64_t dim, int descending);
int lamp_median_dim(lamp_tensor** values, lamp_tensor** indices, const lamp_tensor* a, int64_t dim, int keepdim);   /* median_1: the lower median */
int lamp_unique(lamp_tensor** values, lamp_tensor** inverse_or_null, lamp_tensor** counts_or_null, const lamp_tensor* a);   /* _unique / _unique2, sorted */
/* unique_dim (STen.scala:1059): the distinct slices along `dim`, sorted lexicographically over their flattened elements (ATen's order), the
 * run of every input slice and the multiplicities; unique_consecutive (STen.scala:1068): runs of equal NEIGHBOURING slices, in place order */
int lamp_unique_dim(lamp_tensor** values, lamp_tensor** inverse_or_null, lamp_tensor** counts_or_null, const lamp_tensor* a, int64_t dim);
int lamp_unique_consecutive(lamp_tensor** values, lamp_tensor** inverse_or_null, lamp_tensor** counts_or_null, const lamp_tensor* a, int64_t dim);
/* mode (STen.scala:1561): per slice along dim the smallest most frequent value and the position of its last occurrence */
int lamp_mode(lamp_tensor** values, lamp_tensor** indices, const lamp_tensor* a, int64_t dim, int keepdim);
/* cartesian_prod (STen.scala:674): rows = all combinations of the 1-D tensors' elements, first tensor slowest */
int lamp_cartesian_prod(lamp_tensor** out, lamp_tensor* const* tensors, int n);
int lamp_bincount(lamp_tensor** out, const lamp_tensor* a, const lamp_tensor* weights_or_null, int64_t minlength);
/* Overwriting scatters (STen.scala:1412-1423 scatter, :1715-1726 indexPut / put / indexCopy).  Duplicate targets without accumulation: one
 * of the writers wins (unspecified in ATen too); out-of-range indices raise at the next host wait (device assertion, like ATen's). */
int lamp_scatter(lamp_tensor** out, const lamp_tensor* self, int64_t dim, const lamp_tensor* index, const lamp_tensor* src);          /* scatter_0 */
int lamp_scatter_value(lamp_tensor** out, const lamp_tensor* self, int64_t dim, const lamp_tensor* index, double value);              /* scatter_1 */
int lamp_index_put(lamp_tensor** out, const lamp_tensor* self, lamp_tensor* const* indices, int n, const lamp_tensor* values, int accumulate);
int lamp_put(lamp_tensor** out, const lamp_tensor* self, const lamp_tensor* index, const lamp_tensor* values, int accumulate);
int lamp_index_copy(lamp_tensor** out, const lamp_tensor* self, int64_t dim, const lamp_tensor* index, const lamp_tensor* source);
/* Triangles and diagonals (STen.scala:1883-1886, :1322) */
int lamp_tril(lamp_tensor** out, const lamp_tensor* a, int64_t diagonal);
int lamp_triu(lamp_tensor** out, const lamp_tensor* a, int64_t diagonal);
int lamp_tril_out(lamp_tensor* out, const lamp_tensor* a, int64_t diagonal);           /* out may be a itself (STen.tril_) */
int lamp_diagonal(lamp_tensor** out, const lamp_tensor* a, int64_t offset, int64_t dim1, int64_t dim2);   /* a view */
int lamp_trace(lamp_tensor** out, const lamp_tensor* a);
int lamp_one_hot(lamp_tensor** out, const lamp_tensor* a, int64_t num_classes);
int lamp_embedding(lamp_tensor** out, const lamp_tensor* weight, const lamp_tensor* indices);
/* ATen embedding_backward(grad, indices, num_weights, padding_idx, false, false): rows equal to padding_idx get no gradient;
 * lamp always passes padding_idx = 0 (ops.scala:2150-2158) */
int lamp_embedding_backward(lamp_tensor** out, const lamp_tensor* grad, const lamp_tensor* indices, int64_t num_weights, int64_t padding_idx);
/* RNG: Philox4x32-10 counter streams; bit-compat with libtorch streams is not required by any
 * reference test (SURVEY.md 8b "RNG / globals") */
int lamp_rand(lamp_tensor** out, const int64_t* sizes, int ndim, int dtype, int device);
int lamp_randn(lamp_tensor** out, const int64_t* sizes, int ndim, int dtype, int device);
int lamp_normal(lamp_tensor** out, double mean, double std, const int64_t* sizes, int ndim, int dtype, int device);
int lamp_randint(lamp_tensor** out, int64_t low, int64_t high, const int64_t* sizes, int ndim, int dtype, int device);
/* STen.randperm (:274, int64) and STen.multinomial (:259-264; the language model's sampler, languagemodel/package.scala:100): the library's
 * Philox stream; with replacement an inverse-CDF draw per sample, without replacement the num_samples first of the exponential clocks E / p. */
int lamp_randperm(lamp_tensor** out, int64_t n, int dtype, int device);
int lamp_multinomial(lamp_tensor** out, const lamp_tensor* probs, int64_t num_samples, int replacement);
int lamp_dropout_(lamp_tensor* self, double p, int training);

/* ------------------------------------------------------------------------------------------
 * fused multi-tensor optimiser steps   (lamp-core/src/main/scala/lamp/nn/AdamW.scala:101-176,
 * SGD.scala:46-98, nn/package.scala:72-100).  One launch for all parameter tensors instead
 * of ~8 ATen calls per tensor; arithmetic order follows the reference line by line.
 * ------------------------------------------------------------------------------------------ */
/* sum_i ||g_i||^2 -> out (1 element, dtype of g_0); then g_i *= min(1, theta/sqrt(sum)) */
int lamp_gradient_clipping_(lamp_tensor* const* grads, int n, double theta);
/* params/grads/m/v: n tensors each. master_or_null[i] != NULL => mixed precision working copy
 * (f32) is updated and params[i] receives the down-cast copy (AdamW.scala:48-85,167-169). */
int lamp_adamw_step_(lamp_tensor* const* params, lamp_tensor* const* grads, lamp_tensor* const* m,
                     lamp_tensor* const* v, lamp_tensor* const* master_or_null, int n, const double* lr,
                     const double* weight_decay, const double* beta1, const double* beta2, double eps,
                     double schedule_factor, int64_t step_count, int debias);
int lamp_sgdw_step_(lamp_tensor* const* params, lamp_tensor* const* grads, lamp_tensor* const* velocity_or_null,
                    int n, const double* lr, const double* weight_decay, const double* momentum,
                    double schedule_factor);
/* flat bucket helpers for data parallel training: pack n tensors into one
 * contiguous f32 bucket scaled by `scale`, and unpack with a (device-resident) divisor */
int lamp_flatten_into_(lamp_tensor* bucket, lamp_tensor* const* ts, int n, double scale);
int lamp_unflatten_from_(lamp_tensor* const* ts, int n, const lamp_tensor* bucket, int divide_by_last_element);

/* ------------------------------------------------------------------------------------------
 * scaled dot product attention   (aten _scaled_dot_product_*_attention(+_backward):
 * STen.scala:501-584; ops.scala:2342-2390).  q,k,v: (B, heads, S, d).
 * ------------------------------------------------------------------------------------------ */
int lamp_scaled_dot_product_attention(lamp_tensor** out, lamp_tensor** logsumexp, const lamp_tensor* q,
                                      const lamp_tensor* k, const lamp_tensor* v, int is_causal, double scale);
int lamp_scaled_dot_product_attention_backward(lamp_tensor* out3[3], const lamp_tensor* grad_out,
                                               const lamp_tensor* q, const lamp_tensor* k, const lamp_tensor* v,
                                               const lamp_tensor* out, const lamp_tensor* logsumexp,
                                               int is_causal, double scale);
/* the same with ScaledDotProductAttention's optional additive `attentionBias` (ops.scala:2342-2390; it receives no gradient): broadcasts
 * against (B, heads, Sq, Sk); NULL = none */
int lamp_scaled_dot_product_attention_bias(lamp_tensor** out, lamp_tensor** logsumexp, const lamp_tensor* q, const lamp_tensor* k,
                                           const lamp_tensor* v, const lamp_tensor* attn_bias_or_null, int is_causal, double scale);
int lamp_scaled_dot_product_attention_bias_backward(lamp_tensor* out3[3], const lamp_tensor* grad_out, const lamp_tensor* q,
                                                    const lamp_tensor* k, const lamp_tensor* v, const lamp_tensor* out,
                                                    const lamp_tensor* logsumexp, const lamp_tensor* attn_bias_or_null, int is_causal,
                                                    double scale);

/* ------------------------------------------------------------------------------------------
 * kNN / UMAP fused kernels   (lamp-knn/src/main/scala/lamp/knn/package.scala:21-80;
 * lamp-umap/src/main/scala/lamp/umap/umap.scala:115-286)
 * ------------------------------------------------------------------------------------------ */
/* indices[q,k] (i64) of the k smallest max(0,|q|^2+|x|^2-2q.x) per query row; never
 * materialises the q x n distance matrix.  distances_or_null receives the k values. */
int lamp_knn_squared_euclidean(lamp_tensor** indices, lamp_tensor** distances_or_null, const lamp_tensor* data,
                               const lamp_tensor* query, int64_t k);
/* the same search under lamp.knn.JaccardDistance (package.scala:16-19, 32-44): 1 - q.x / ((sum q + sum x) - q.x) */
int lamp_knn_jaccard(lamp_tensor** indices, lamp_tensor** distances_or_null, const lamp_tensor* data,
                     const lamp_tensor* query, int64_t k);
/* one fused evaluation of the UMAP layout loss (umap.scala:132-176) and of its gradient w.r.t.
 * `locations`, accumulated into grad_accum:
 *   loss = -(attractions / sum(b) + repulsions * repulsion_strength / |index3|)        (balance != 0)
 *   grad_accum += sum_k term_weights[k] * scatter_k(dloss/dlocations_k),  k = index1..index4.
 * term_weights reproduces how lamp's IndexSelect backward (`out += out.indexAdd(..)`, ops.scala:186-191)
 * accumulates the four gathers into one buffer: {1, 2, 4, 8} in the reference's traversal order;
 * pass {1, 1, 1, 1} for the mathematical gradient. */
/* lamp_knn_squared_euclidean (knnSearch with SquaredEuclideanDistance, knn/package.scala:60-121) on large f32 / f64 searches (64 / 128
 * features, k <= 12) runs a filter on the 16-bit matrix pipe (every value split exactly into two f16 pieces, three products per feature)
 * that keeps 16 candidates per query, re-ranks them with exact distances in the data's precision, and PROVES per query that no other
 * point can be among the k nearest; queries without proof go through the exact kernel.  The result is that of the exact search
 * (kernels/knn_split.hip).  mode 0: never, 1: where it pays (default), 2: whenever the shape is covered.  Process-wide; no counterpart
 * in the reference (its ATen call has one implementation). */
int lamp_knn_split_mode(int mode);
int lamp_knn_split_last_failed(int64_t* out);   /* queries of the last split search that needed the exact kernel */
int lamp_knn_split_last_planes(int* out);       /* f16 planes per value it used (2; 0: the split path did not run) */
/* knnDistances of Umap.umap (umap.scala:382-402, a JVM loop in the reference): out[i, j] = exact f64 Euclidean distance
 * between data row i and data row indices[i, j], summed left to right over the columns.  out is [n, k] f64. */
int lamp_knn_row_distances(lamp_tensor** out, const lamp_tensor* data, const lamp_tensor* indices);
/* Umap.edgeWeights (umap.scala:50-113, JVM double loops in the reference): per point the smallest positive kNN
 * distance rho and the bisection for sigma (Umap.binarySearch, umap.scala:14-48), then for every neighbour j != i
 * the fuzzy union b = w_ij + w_ji - w_ij*w_ji.  knn_distances [n,k] f64, knn [n,k] i64 -> out [m,3] f64 rows (i, j, b) in
 * the reference's emission order. */
int lamp_umap_edge_weights(lamp_tensor** out, const lamp_tensor* knn_distances, const lamp_tensor* knn);
int lamp_umap_loss_grad(lamp_tensor** loss, lamp_tensor* grad_accum, const lamp_tensor* locations,
                        const lamp_tensor* index1, const lamp_tensor* index2, const lamp_tensor* b,
                        const lamp_tensor* index3, const lamp_tensor* index4, double min_dist,
                        int balance, double repulsion_strength, const double* term_weights);
/* Same, with the reference's `mask = i.ne(j); i.maskedSelect(mask); j.maskedSelect(mask)` (umap.scala:221-227) folded in:
 * negative pairs with index3[e] == index4[e] are ignored and the repulsion is normalised by the number of pairs kept
 * (counted on the device).  Saves the two 45M-element compactions per iteration at 1M points. */
int lamp_umap_loss_grad_skip_self(lamp_tensor** loss, lamp_tensor* grad_accum, const lamp_tensor* locations,
                                  const lamp_tensor* index1, const lamp_tensor* index2, const lamp_tensor* b,
                                  const lamp_tensor* index3, const lamp_tensor* index4, double min_dist,
                                  int balance, double repulsion_strength, const double* term_weights);
/* Umap.optimize's negatives drawn by the library (round 6).  lamp_umap_negatives: ii = index1.repeatInterleave(n), jj = randint(0, high, [E1 * n])
 * (umap.scala:211-213) from one counter block of the generator.  lamp_umap_loss_grad_sampled = lamp_umap_loss_grad_skip_self on exactly those
 * negatives when it takes the generator at the same point (lamp_manual_seed + the same calls since) - drawn INSIDE the kernels for 2-D layouts: the
 * per-iteration index tensors (0.72 GB at 1M points), repeatInterleave, randint and the pair count's pass over them do not exist. */
int lamp_umap_negatives(lamp_tensor** ii, lamp_tensor** jj, const lamp_tensor* index1, int64_t negatives_per_edge, int64_t high);
int lamp_umap_loss_grad_sampled(lamp_tensor** loss, lamp_tensor* grad_accum, const lamp_tensor* locations, const lamp_tensor* index1,
                                const lamp_tensor* index2, const lamp_tensor* b, int64_t negatives_per_edge, int64_t high, double min_dist,
                                int balance, double repulsion_strength, const double* term_weights);

/* The layout with its EDGE LIST sharded over ranks (lamp-umap is single-device): every rank evaluates its slice of the
 * attractive edges and its own negatives under GLOBAL normalisers - bsum (sum of b over all ranks, one element of b's dtype) and
 * kept (all-reduced lamp_count_ne of the negatives, one int64) - and the caller all-reduces gradient and loss. */
int lamp_count_ne(lamp_tensor** out /* int64 [1] on the device */, const lamp_tensor* a, const lamp_tensor* b);
int lamp_umap_loss_grad_sharded(lamp_tensor** loss, lamp_tensor* grad_accum, const lamp_tensor* locations,
                                const lamp_tensor* index1, const lamp_tensor* index2, const lamp_tensor* b,
                                const lamp_tensor* index3, const lamp_tensor* index4, double min_dist,
                                int balance, double repulsion_strength, const double* term_weights,
                                const lamp_tensor* bsum_global, const lamp_tensor* kept_global);

/* ------------------------------------------------------------------------------------------
 * collectives over RCCL / xGMI  (aten.NcclComm.{get_unique_id, comm_init_rank, broadcast,
 * reduce, comm_destroy}: STen.scala:629-671,1902-1908; call sites
 * lamp-data/.../distributed/package.scala:683-731).  all_reduce is the addition the
 * data-parallel redesign needs (one flat bucket instead of 74 broadcasts + 38 reduces).
 * ------------------------------------------------------------------------------------------ */
#define LAMP_UNIQUE_ID_BYTES 128
int lamp_comm_get_unique_id(uint8_t* id_out /* LAMP_UNIQUE_ID_BYTES */);
int lamp_comm_init_rank(lamp_comm** out, int nranks, const uint8_t* id, int rank);
int lamp_comm_broadcast(lamp_tensor* const* tensors, lamp_comm* const* comms, int n, int root);
int lamp_comm_reduce(lamp_tensor* const* inputs, lamp_tensor* output, int root, int op /* ncclRedOp_t: 0 sum, 1 prod, 2 max, 3 min */,
                     lamp_comm* const* comms, int n);
int lamp_comm_all_reduce(lamp_tensor* const* tensors, lamp_comm* const* comms, int n, int op);
/* ncclCommCount / ncclCommUserRank: what RCCL itself says about the clique (bench.py prints rccl_ranks from this and refuses
 * to report an N-GPU figure unless it equals N) */
int lamp_comm_count(const lamp_comm* c, int* nranks_out);
int lamp_comm_user_rank(const lamp_comm* c, int* rank_out);
/* out = concatenation over ranks of `in` (ncclAllGather): the exchange step of row-sharded work, e.g. the kNN graph whose
 * query rows are split over the ranks (lamp-knn has no multi-GPU form) */
int lamp_comm_all_gather(lamp_tensor* out, const lamp_tensor* in, lamp_comm* comm);
int lamp_comm_destroy(lamp_comm* c);

#ifdef __cplusplus
}
#endif
#endif /* LAMP_HIP_H */
