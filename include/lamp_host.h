/*
 * lamp_host.h - C ABI of the host-side mirror of lamp-core's autograd / nn / optimiser layer.
 *
 * In the reference this layer is Scala running on the JVM ABOVE the JNI boundary
 * (lamp-core/src/main/scala/lamp/autograd/{autograd,ops}.scala, lamp/nn (all files),
 * lamp-data/.../distributed/package.scala); it only sequences aten.* calls.  No JVM exists in
 * the build environment, so the same sequencing logic is restated in C++ over lamp_hip.h and
 * exported here so that the parity tests and bench.py can drive the exact op sequences the
 * reference would issue.  A JVM deployment does not need this header: lamp-core itself runs
 * unchanged on top of lamp_hip.h.
 *
 * Same conventions as lamp_hip.h: int status, lamp_last_error(), out-params own +1 handle.
 */
#ifndef LAMP_HOST_H
#define LAMP_HOST_H

#include "lamp_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lamp_var lamp_var;             /* lamp.autograd.Variable (autograd.scala:176) */
typedef struct lamp_module lamp_module;       /* lamp.nn.GenericModule (Module.scala:272) */
typedef struct lamp_optimizer lamp_optimizer; /* lamp.nn.Optimizer (Optimizer.scala:5) */
typedef struct lamp_model lamp_model;         /* lamp.nn.SupervisedModel (SupervisedModel.scala:151) */

/* ---- variables (autograd/package.scala:60-78; autograd.scala:176-282) ---- */
int lamp_var_const(lamp_var** out, const lamp_tensor* value);
int lamp_var_param(lamp_var** out, const lamp_tensor* value);
int lamp_var_value(const lamp_var* v, lamp_tensor** out);
int lamp_var_grad(const lamp_var* v, lamp_tensor** out);          /* *out = NULL when needsGrad is false */
int lamp_var_needs_grad(const lamp_var* v, int* out);
int lamp_var_zero_grad(lamp_var* v);
int lamp_var_backprop(lamp_var* v);                               /* Variable.backprop */
int lamp_var_wengert_size(const lamp_var* v, int64_t* out);       /* length of the topological order */
int lamp_var_release(lamp_var* v);

/* One entry point for every Op case class of ops.scala, selected by its Scala name
 * ("MatMul", "Add", "Relu", "Convolution", "BatchNorm2D", "NllLoss", ...):
 *   vars    - Variable inputs in the order of the case class parameters (NULL for absent options)
 *   tensors - plain tensor arguments (targets, weights, running statistics)
 *   d / i   - double / long scalar arguments in declaration order */
int lamp_op_apply(lamp_var** out, const char* name, lamp_var* const* vars, int nvars, lamp_tensor* const* tensors, int ntensors,
                  const double* d, int nd, const int64_t* i, int ni);

/* ---- modules (lamp/nn) ---- */
int lamp_module_linear(lamp_module** out, int64_t in, int64_t outf, int dtype, int device, int bias);
int lamp_module_conv2d(lamp_module** out, int64_t in_channels, int64_t out_channels, int64_t kernel, int dtype, int device, int bias,
                       int64_t stride, int64_t padding, int64_t dilation, int64_t groups);
int lamp_module_batch_norm(lamp_module** out, int64_t features, int dtype, int device, int two_d);
int lamp_module_layer_norm(lamp_module** out, const int64_t* shape, int nshape, int dtype, int device, int scale, int bias);
int lamp_module_dropout(lamp_module** out, double p);
int lamp_module_fun(lamp_module** out, const char* name, double a, double b);
int lamp_module_sequential(lamp_module** out, lamp_module* const* mods, int n);
int lamp_module_residual(lamp_module** out, lamp_module* right, lamp_module* left_or_null);
int lamp_module_mlp(lamp_module** out, int64_t in, int64_t outf, const int64_t* hidden, int nhidden, int dtype, int device, double dropout,
                    int last_non_linearity, const char* activation, int norm, int bias);
/* Cnn.resnet (example-cifar100/src/main/scala/lamp/example/cifar/cnn.scala:89-137) */
int lamp_module_resnet(lamp_module** out, int64_t num_classes, double dropout, int dtype, int device);
/* ---- transformer family and the autoregressive language model (lamp-core nn/Transformer.scala, nn/Embedding.scala,
 * nn/languagemodel/lm.scala).  State order = the reference's `state`; the layer norms inside the blocks carry no scale / bias
 * (LayerNorm.apply defaults).  Modules whose input is a tuple / case class are called through lamp_module_forward_multi. */
int lamp_module_embedding(lamp_module** out, int64_t classes, int64_t dimensions, int dtype, int device);            /* Embedding.scala:35-48 */
int lamp_module_multihead_attention(lamp_module** out, int64_t dQ, int64_t dK, int64_t dV, int64_t hidden_per_head, int64_t outf, double dropout,
                                    int64_t num_heads, int dtype, int device, int linearized, int causal_mask);      /* Transformer.scala:619-641; vars (q, k, v), tensors (maxLength?) */
/* What the fused-operator branch of MultiheadAttention.multiheadAttention (Transformer.scala:946-962) computes.  0 (default): the
 * arithmetic of the reference's ATen CPU path (its composed branch, :963-1001): per-head attention over the sequence; the flash
 * kernels read (batch, heads, sequence, d) views of the projections in place.  1: the call as written for CUDA - (batch, sequence,
 * heads, d) views handed to an operator that reads dimension 1 as heads (attention over the heads of each token).  Process-wide;
 * returns the previous value in *previous_or_null.  Environment default: LAMP_ATTENTION_AS_WRITTEN_FOR_CUDA=1. */
int lamp_attention_fused_call_as_written(int on, int* previous_or_null);
int lamp_module_transformer_encoder_block(lamp_module** out, int64_t in, int64_t attention_hidden_per_head, int64_t attention_num_heads,
                                          int64_t mlp_hidden, int64_t outf, double dropout, int dtype, int device, int linearized, int gpt_order,
                                          int causal_mask);                                                            /* :492-530; vars (x), tensors (maxLength?) */
int lamp_module_transformer_encoder(lamp_module** out, int64_t num_blocks, int64_t in, int64_t attention_hidden_per_head, int64_t attention_num_heads,
                                    int64_t mlp_hidden, double dropout, int dtype, int device, int linearized, int gpt_order, int causal_mask); /* :76-102 */
int lamp_module_transformer_decoder_block(lamp_module** out, int64_t in, int64_t attention_hidden_per_head, int64_t attention_num_heads,
                                          int64_t mlp_hidden, int64_t outf, double dropout, int dtype, int device, int linearized,
                                          int decoder_decoder_causal_mask, int encoder_decoder_causal_mask);          /* :384-432; vars (decoderInput, encoderOutput), tensors (maxLength?) */
int lamp_module_transformer(lamp_module** out, int64_t num_blocks, int64_t in, int64_t attention_hidden_per_head, int64_t attention_num_heads,
                            int64_t mlp_hidden, double dropout, int dtype, int device, int linearized, int encoder_causal_mask,
                            int decoder_decoder_causal_mask, int encoder_decoder_causal_mask); /* :330-365; vars (decoderInput, encoderInput), tensors (decoderMaxLength?, encoderMaxLength?) */
int lamp_positional_embedding_vaswani(lamp_tensor** out, int64_t sequence_length, int64_t dimension, int dtype, int device); /* :1022-1045 */
int lamp_module_transformer_embedding(lamp_module** out, lamp_module* embedding, int add_positional_embedding,
                                      const lamp_tensor* positional_embedding);                                        /* :1105-1125 */
int lamp_module_language_model(lamp_module** out, int64_t max_length, int64_t vocabulary_size, int64_t num_blocks, int64_t embedding_dim,
                               int64_t attention_hidden_per_head, int64_t attention_num_heads, int64_t encoder_mlp_hidden, double dropout, int dtype,
                               int device, int linearized);                      /* lm.scala:194-232; vars (tokens), tensors (maxLength?, positions?) -> logits */
int lamp_module_language_model_loss(lamp_module** out, int64_t max_length, int64_t vocabulary_size, int64_t num_blocks, int64_t embedding_dim,
                                    int64_t attention_hidden_per_head, int64_t attention_num_heads, int64_t encoder_mlp_hidden, double dropout,
                                    int64_t pad_token, int dtype, int device, int linearized); /* lm.scala:63-91; vars (tokens), tensors (target, maxLength?, positions?) -> loss */
/* LanguageModelOutput(encoded, languageModelLogits) (lm.scala:146-188); either output pointer may be NULL */
int lamp_language_model_forward(lamp_module* m, lamp_var* tokens, const lamp_tensor* max_length_or_null, const lamp_tensor* positions_or_null,
                                lamp_var** encoded, lamp_var** logits);
int lamp_sequence_mask(lamp_var** out, const lamp_tensor* max_length, lamp_var* maskable, double fill);   /* MultiheadAttention.sequenceMask :667-749 */
int lamp_masked_softmax(lamp_var** out, lamp_var* input, const lamp_tensor* max_length);                  /* MultiheadAttention.maskedSoftmax :751-762 */
/* GenericModule[A, B].forward for a tuple / case-class A: its Variables and its plain tensors in the reference's order; a NULL tensor = None */
int lamp_module_forward_multi(lamp_module* m, lamp_var* const* vars, int nvars, const lamp_tensor* const* tensors, int ntensors, lamp_var** out);
int lamp_module_forward(lamp_module* m, lamp_var* x, lamp_var** out);
int lamp_module_num_state(lamp_module* m, int64_t* out);
int lamp_module_state(lamp_module* m, int64_t index, lamp_var** out);   /* state in lamp's order (params and consts) */
int lamp_module_set_training(lamp_module* m, int training);              /* asTraining / asEval */
int lamp_module_zero_grad(lamp_module* m);
int lamp_module_release(lamp_module* m);

/* ---- optimisers (nn/AdamW.scala, nn/SGD.scala); clip < 0 means None ---- */
int lamp_optimizer_adamw(lamp_optimizer** out, lamp_tensor* const* params, int n, double weight_decay, double learning_rate, double beta1,
                         double beta2, double eps, double clip, int debias, int mixed_precision);
/* AdamW with OptimizedHyperparameter = PTag => Double resolved per parameter (AdamW.scala:29-47, train.scala:48-66 of
 * example-autoregressivelm: weight decay on the attention / MLP matrices only): arrays of n values */
int lamp_optimizer_adamw_tagged(lamp_optimizer** out, lamp_tensor* const* params, int n, const double* weight_decay, const double* learning_rate,
                                const double* beta1, const double* beta2, double eps, double clip, int debias, int mixed_precision);
int lamp_optimizer_sgdw(lamp_optimizer** out, lamp_tensor* const* params, int n, double learning_rate, double weight_decay,
                        double momentum /* < 0: none */, double clip);
int lamp_optimizer_step(lamp_optimizer* o, lamp_tensor* const* gradients /* NULL entries = None */, int n, double schedule_factor);
int lamp_optimizer_num_state(lamp_optimizer* o, int64_t* out);
int lamp_optimizer_state(lamp_optimizer* o, int64_t index, lamp_tensor** out);
/* Optimizer.load (AdamW.scala:87-93, SGD.scala:38-42): copyFrom into the state tensors in order; AdamW re-reads its step count */
int lamp_optimizer_load(lamp_optimizer* o, lamp_tensor* const* tensors, int n);
int lamp_optimizer_release(lamp_optimizer* o);
/* nn/package.scala:72-100 */
int lamp_gradient_clipping_in_place(lamp_tensor* const* gradients, int n, double theta);

/* ---- supervised model + training steps ---- */
/* loss_kind 0: LossFunctions.NLL(numClasses, classWeights, reduction, ignore) ; 1: MSE ; 2: Identity */
int lamp_model_create(lamp_model** out, lamp_module* module, int loss_kind, const lamp_tensor* class_weights_or_null, int64_t reduction,
                      int64_t ignore_index);
/* SupervisedModel.addTotalLossAndReturnGradientsAndNumExamples (SupervisedModel.scala:190-211):
 * forward, loss, zero grads, backprop; acc += loss * n. Gradients are left in the parameters' grad buffers. */
int lamp_model_gradients(lamp_model* m, const lamp_tensor* samples, const lamp_tensor* target, lamp_tensor* acc_or_null, int zero_grad,
                         int64_t* num_examples);
int lamp_model_forward_loss(lamp_model* m, const lamp_tensor* samples, const lamp_tensor* target, lamp_tensor* acc_or_null,
                            int64_t* num_examples);
/* one full step = gradients + (optional RCCL exchange) + optimizer.step, as IOLoops.oneEpoch's processBatch
 * (lamp-data/.../IOLoops.scala:621-658) or distributed oneBatch (distributed/package.scala:733-759) */
int lamp_model_train_step(lamp_model* m, lamp_optimizer* o, lamp_comm* comm_or_null, const lamp_tensor* samples, const lamp_tensor* target,
                          lamp_tensor* acc_or_null, int64_t* num_examples);
/* the same with the learning-rate schedule factor of the epoch (IOLoops.scala:621-658 `learningRateScheduleFactor`) */
int lamp_model_train_step_scheduled(lamp_model* m, lamp_optimizer* o, lamp_comm* comm_or_null, const lamp_tensor* samples,
                                    const lamp_tensor* target, lamp_tensor* acc_or_null, double schedule_factor, int64_t* num_examples);
/* broadcast of module.state (parameters and batch-norm running statistics) and the optimiser state from `root` to every rank
 * (distributed/package.scala:683-688).  lamp_model_train_step does it by itself before the first step over a communicator; call it
 * again before validation / checkpoints to make rank `root`'s non-parameter state the state of every rank. */
int lamp_model_sync_state(lamp_model* m, lamp_optimizer* o, lamp_comm* comm, int root);
/* averageGradients + optimizer.step (distributed/package.scala:690-759) on gradients the caller has already computed - e.g. by replaying
 * forward + backprop from a HIP graph: g_i *= n, one flat all-reduce with n, g_i /= sum n, step.  The replicas are made identical first
 * if they have not been over this communicator. */
int lamp_model_exchange_and_step(lamp_model* m, lamp_optimizer* o, lamp_comm* comm, lamp_tensor* const* grads, int ngrads, int64_t num_examples,
                                 double schedule_factor);
/* Single-process data parallel, DataParallel.driveSynchronousLoop's `synchronousStep` (lamp-data DataParallel.scala:195-311): the main
 * model (with the optimiser) and `nreplicas` replicas on other GPUs; arrays of nreplicas + 1 entries, main first.  Copies the main
 * state to the replicas, computes every model's gradients on its own host thread, and - when `step` - averages the gradients weighted
 * by the example counts into the main model and steps the optimiser.  Returns the examples of all models. */
int lamp_data_parallel_step(lamp_model* main_model, lamp_optimizer* o, lamp_model* const* replicas, int nreplicas, const lamp_tensor* const* samples,
                            const lamp_tensor* const* targets, lamp_tensor* const* accs_or_null, int zero_grad, int step, double schedule_factor,
                            int64_t* num_examples);
int lamp_model_release(lamp_model* m);

/* ---- tensor-list files and checkpoints (lamp-data Writer.scala:14-190, Reader.scala:17-95, schemas.scala:30-56) ----
 * A list of tensors is a JSON descriptor {"tensors":[{"dims","dataType","byteOffset","byteLength"}...],"location",
 * "byteOffset","byteLength"} at `path` plus the raw little-endian blob `path`.data (every tensor padded to a multiple
 * of 8 bytes); `location` is relative to the descriptor unless absolute.  Files written here are readable by the
 * reference's Reader and the other way round. */
int lamp_write_tensors_into_file(lamp_tensor* const* tensors, int64_t n, const char* path);    /* Writer.writeTensorsIntoFile */
int lamp_tensor_list_length(const char* path, int64_t* n);
/* Reader.readTensorsFromFile(file, device, pin) with STen.tensorsFromFile's checks (STen.scala:157-168); device -1 = host */
int lamp_read_tensors_from_file(lamp_tensor** out /* capacity */, int64_t capacity, int64_t* n_read, const char* path, int device, int pin);
int lamp_module_write_checkpoint(lamp_module* m, const char* path);     /* Writer.writeCheckpoint: module.state in order */
int lamp_module_load_from_file(lamp_module* m, const char* path);       /* Reader.loadFromFile: copyFrom per state tensor */

/* Cifar.loadImageFile (example-cifar100/.../cifar100.scala:29-56): records of 3074 bytes (coarse label, fine label,
 * 3x32x32 pixels); labels = i64 fine labels [n], images = [n, 3, 32, 32] cast to `dtype` (raw 0..255 values). */
int lamp_cifar_load_image_file(lamp_tensor** labels, lamp_tensor** images, const char* path, int64_t num_images, int dtype, int device);

/* ---- BatchStream.minibatchesFromFull (BatchStream.scala:528-592) + everyNth (:378-400) ----
 * `order` is the shuffled row order (the caller owns the RNG: scala.util.Random.shuffle on the JVM); it is cut into groups of
 * minibatch_size; drop_last removes the LAST group whether or not it is full, as the reference does.  The data set lives in
 * HBM and a minibatch is one gather on the device.  next: *x = NULL at EndStream. */
typedef struct lamp_batch_stream lamp_batch_stream;
int lamp_batch_stream_from_full(lamp_batch_stream** out, const lamp_tensor* features, const lamp_tensor* target, const int64_t* order, int64_t n,
                                int64_t minibatch_size, int drop_last, int device);
/* The same stream over a data set that STAYS IN HOST MEMORY, as in the reference (BatchStream.scala:539-556: host gather, pinned staging buffer,
 * copy on another stream; `pinned` of cifar100.scala): the features are pinned (copied once if they are not), and the GPU gathers a
 * minibatch's rows over PCIe, queued one batch ahead of the consumer (IOLoops.scala:833-874), converting to out_dtype (-1: as
 * stored) on the way.  Batches, order and values are those of lamp_batch_stream_from_full. */
int lamp_batch_stream_from_full_host(lamp_batch_stream** out, const lamp_tensor* features, const lamp_tensor* target, const int64_t* order, int64_t n,
                                     int64_t minibatch_size, int drop_last, int device, int out_dtype);
int lamp_batch_stream_every_nth(lamp_batch_stream* s, int64_t n, int64_t offset);
int lamp_batch_stream_num_batches(const lamp_batch_stream* s, int64_t* out);
int lamp_batch_stream_next(lamp_batch_stream* s, lamp_tensor** x, lamp_tensor** target);
int lamp_batch_stream_reset(lamp_batch_stream* s);
int lamp_batch_stream_release(lamp_batch_stream* s);

#ifdef __cplusplus
}
#endif
#endif /* LAMP_HOST_H */
