// lamp's autograd operators (lamp-core/src/main/scala/lamp/autograd/ops.scala) over the C ABI.
#pragma once
#include "autograd.h"

namespace lamp {
namespace host {
namespace F {

Var transpose(const Var& a, int64_t d1 = 0, int64_t d2 = 1);
Var view(const Var& a, const std::vector<int64_t>& shape);
Var reshape(const Var& a, const std::vector<int64_t>& shape);
Var flatten(const Var& a, int64_t start, int64_t end = -1);
Var concatenate(const std::vector<Var>& as, int64_t dim);
Var add(const Var& a, const Var& b);
Var const_add(const Var& a, double b);
Var minus(const Var& a, const Var& b);
Var const_mult(const Var& a, double b);
Var mult(const Var& a, const Var& b);
Var mult_add(const Var& a, const Var& b, const Var& c);   // (a * b) + c in one launch, values of the chain
Var div(const Var& a, const Var& b);
Var sum(const Var& a, const std::vector<int64_t>& dim = {}, bool keepDim = false);
Var mean(const Var& a, const std::vector<int64_t>& dim, bool keepDim = true);
Var norm2(const Var& a, const std::vector<int64_t>& dim, bool keepDim);
Var mm(const Var& a, const Var& b);
Var bmm(const Var& a, const Var& b);
Var linear_bias(const Var& x, const Var& w, const Var& bias);   // x.mm(w) + bias[1, n] in one launch, values of the chain
Var exp(const Var& a);
Var log(const Var& a);
Var log1p(const Var& a);
Var sin(const Var& a);
Var cos(const Var& a);
Var tanh(const Var& a);
Var pow_const(const Var& a, double e);
Var relu(const Var& a);
Var leaky_relu(const Var& a, double slope);
Var gelu(const Var& a);
Var sigmoid(const Var& a);
Var hardswish(const Var& a);
Var softplus(const Var& a, double beta, double threshold);
Var log_softmax(const Var& a, int64_t dim);
Var dropout(const Var& a, double prob, bool train);
Var nll_loss(const Var& input, const Ten& target, const Ten& weights, int64_t reduction = 1, int64_t ignore = -100);
Var nll_loss_accumulate(const Var& input, const Ten& target, const Ten& weights, int64_t reduction, int64_t ignore, const Ten& acc, double scale);
Var mse_loss(const Var& input, const Ten& target, int64_t reduction = 1);
Var index_select(const Var& input, int64_t dim, const Var& index);
Var mask_fill(const Var& input, const Ten& mask, double fill);   // MaskFill (ops.scala:148-159)
Var euclidean_distance(const Var& a, const Var& b, int64_t dim);
Var capped_shifted_negative_exponential(const Var& a, double shift);
Var packed_self_attention(const Var& x, const Var& wq, const Var& wk, const Var& wv, int64_t numHeads, bool isCausal);   // projections + fused attention, one node
Var scaled_dot_product_attention(const Var& query, const Var& key, const Var& value, bool isCausal, const Ten& attentionBias = Ten());
Var convolution(const Var& input, const Var& weight, const Var& bias, const std::vector<int64_t>& stride,
                const std::vector<int64_t>& padding, const std::vector<int64_t>& dilation, bool transposed,
                const std::vector<int64_t>& outputPadding, int64_t groups);
std::pair<Var, Var> convolution_pair(const Var& input, const Var& weight_a, const Var& bias_a, const std::vector<int64_t>& stride_a,
                                     const std::vector<int64_t>& padding_a, const std::vector<int64_t>& dilation_a, const Var& weight_b,
                                     const Var& bias_b, const std::vector<int64_t>& stride_b, const std::vector<int64_t>& padding_b,
                                     const std::vector<int64_t>& dilation_b, int64_t groups);
Var avg_pool2d(const Var& input, int64_t k, int64_t stride, int64_t padding);
Var max_pool2d(const Var& input, int64_t k, int64_t stride, int64_t padding, int64_t dilation);
Var global_avg_pool_log_softmax(const Var& input);   // avg_pool2d(k = H = W) -> flatten -> log_softmax(1) in one launch, values of the chain
Var batch_norm(const Var& input, const Var& weight, const Var& bias, const Ten& runningMean, const Ten& runningVar, bool training,
               double momentum, double eps);
Var batch_norm_2d(const Var& input, const Var& weight, const Var& bias, const Ten& runningMean, const Ten& runningVar, bool training,
                  double momentum, double eps);
bool batch_norm_relu_2d_supported(const Var& input);
Var batch_norm_relu_2d(const Var& input, const Var& weight, const Var& bias, const Ten& runningMean, const Ten& runningVar, bool training,
                       double momentum, double eps);
Var batch_norm_add_relu_2d(const Var& input, const Var& addend, const Var& weight, const Var& bias, const Ten& runningMean,
                           const Ten& runningVar, bool training, double momentum, double eps);
bool conv_of_batch_norm_relu_2d_pays(const Var& input, const Var& weight, int64_t stride, int64_t padding, int64_t dilation, int64_t groups);
Var conv_of_batch_norm_relu_2d(const Var& input, const Var& bnWeight, const Var& bnBias, const Ten& runningMean, const Ten& runningVar, double momentum,
                               double eps, const Var& weight, const Var& bias, const std::vector<int64_t>& stride, const std::vector<int64_t>& padding,
                               const std::vector<int64_t>& dilation, int64_t groups);
Var batch_norm2_add_relu_2d(const Var& input, const Var& weight, const Var& bias, const Ten& runningMean, const Ten& runningVar, double momentum, double eps,
                            const Var& input2, const Var& weight2, const Var& bias2, const Ten& runningMean2, const Ten& runningVar2, double momentum2,
                            double eps2);
Var batch_norm2_add_relu_pool_log_softmax_2d(const Var& input, const Var& weight, const Var& bias, const Ten& runningMean, const Ten& runningVar, double momentum, double eps,
                            const Var& input2, const Var& weight2, const Var& bias2, const Ten& runningMean2, const Ten& runningVar2, double momentum2,
                            double eps2);   // relu(bn(input) + bn2(input2)), training mode, one op
Var layer_norm(const Var& input, const Var& weight /*nullable*/, const Var& bias /*nullable*/, const std::vector<int64_t>& normalizedShape,
               double eps);
Var embedding(const Var& input, const Var& weight);

// ops2.cpp: the rest of the operators the reference's gradient suite exercises (ops.scala line numbers there)
Var stack(const std::vector<Var>& as, int64_t dim);
Var select(const Var& a, int64_t dim, int64_t index);
Var slice(const Var& a, int64_t dim, int64_t start, int64_t end, int64_t step);
Var mask_select(const Var& input, const Var& mask);
Var index_fill(const Var& input, int64_t dim, const Var& index, double fill);
Var where(const Ten& condition, const Var& trueBranch, const Var& falseBranch);
Var assign(const Var& abandon, const Var& keep);
Var cast_to_precision(const Var& a, int dtype);
Var scatter_add(const Var& src, const Var& index, int64_t dim, int64_t maxIndex);
Var index_add(const Var& src, const Var& index, int64_t dim, int64_t maxIndex);
Var index_add_to_target(const Var& target, const Var& src, const Var& index, int64_t dim);
Var repeat_interleave(const Var& self, const Var& repeats, int64_t dim);
Var expand_as(const Var& a, const Ten& as);
Var expand(const Var& a, const std::vector<int64_t>& shape);
Var tan(const Var& a);
Var atan(const Var& a);
Var pow(const Var& a, const Var& exponent);
Var minimum(const Var& a, const Var& b);
Var maximum(const Var& a, const Var& b);
Var variance(const Var& a, const std::vector<int64_t>& dim);
Var squared_frobenius(const Var& a);
Var weight_norm(const Var& v, const Var& g, int64_t dim);
Var smooth_l1_loss(const Var& input, const Ten& target, int64_t reduction, double beta);
Var binary_cross_entropy_with_logits(const Var& input, const Ten& target, const Ten& posWeights /* may be undefined */, int64_t reduction);
Var max_pool1d(const Var& input, int64_t k, int64_t stride, int64_t padding, int64_t dilation);
Var diag(const Var& a, int64_t diagonal);
Var cross(const Var& a, const Var& b, int64_t dim);
Var argmax(const Var& a, int64_t dim, bool keepDim);      // not differentiable: backprop through it raises, as in the reference
Var one_hot(const Var& a, int64_t numClasses);            // not differentiable
Var eq_where(const Var& a, int64_t b);

}  // namespace F
}  // namespace host
}  // namespace lamp
