// lamp.nn's transformer family and the autoregressive language model over the C ABI.
//
// Reference: lamp-core/src/main/scala/lamp/nn/Transformer.scala (MultiheadAttention :572-1008, TransformerEncoderBlock :212-260 / :489-570,
// TransformerEncoder :29-103, TransformerDecoderBlock :263-307 / :381-487, TransformerDecoder :105-198, Transformer :310-379,
// PositionalEmbedding.vaswani :1022-1045, TransformerEmbedding :1105-1141), nn/Embedding.scala:17-48,
// nn/languagemodel/lm.scala (LanguageModelModule :137-190, LanguageModelLoss :44-103).
//
// Inputs that the reference passes as tuples arrive through Module::forward_multi(vars, aux tensors); a missing Option is an
// undefined Ten.  Everything here is a composition of the autograd operators in ops.h - the arithmetic runs in the HIP library.
#pragma once
#include "nn.h"

namespace lamp {
namespace host {

struct Embedding : Module {                 // nn/Embedding.scala:17-48 (init N(0, sqrt(2 / (classes + dimensions))))
  Var weights;
  static std::shared_ptr<Embedding> make(int64_t classes, int64_t dimensions, int dtype, int device);
  void collect_state(std::vector<Var>& o) override { o.push_back(weights); }
  Var forward(const Var& x) override { return F::embedding(x, weights); }
};

struct MultiheadAttention : Module {        // Transformer.scala:572-616; state wQ, wK, wV, wO
  Var wQ, wK, wV, wO;
  double dropout = 0; bool train = true; int64_t numHeads = 1; bool linearized = false, causalMask = false;
  static std::shared_ptr<MultiheadAttention> make(int64_t dQ, int64_t dK, int64_t dV, int64_t hiddenPerHead, int64_t out, double dropout,
                                                  int64_t numHeads, int dtype, int device, bool linearized, bool causalMask);
  void collect_state(std::vector<Var>& o) override { o.push_back(wQ); o.push_back(wK); o.push_back(wV); o.push_back(wO); }
  Var attend(const Var& q, const Var& k, const Var& v, const Ten& maxLength);
  Var forward(const Var& x) override { return attend(x, x, x, Ten()); }
  // vars = (query, keys, values) or (x); aux = (maxLength?)
  Var forward_multi(const std::vector<Var>& xs, const std::vector<Ten>& aux) override;
  void set_training(bool) override {}       // TrainingMode.identity (Transformer.scala:644-645)

  // Which arithmetic the fused-operator branch of multiheadAttention (Transformer.scala:946-962) stands for:
  //   false (default) - the meaning of the reference's ATen CPU path, i.e. of its composed branch (:963-1001): every head attends
  //                     over the sequence.  The fused kernels get (batch, heads, sequence, d) VIEWS of the projections;
  //   true            - the call exactly as written for CUDA: (batch, sequence, heads, d) views handed to an operator that reads
  //                     dimension 1 as the heads - attention over the HEADS of each token (SURVEY 8a-17's layout hazard).
  static bool& fused_call_as_written();
  // companion object (Transformer.scala:667-1008)
  static Var sequenceMask(const Ten& maxLength, const Var& maskable, double fill);
  static Var maskedSoftmax(const Var& input, const Ten& maxLength);
  static Var scaledDotProductAttention(const Var& q, const Var& k, const Var& v, const Ten& maxLength, double dropout, bool trainDropout);
  static Var linearizedAttention(const Var& q, const Var& k, const Var& v, const Ten& maxLength, double dropout, bool trainDropout);
  static Var multiheadAttention(const Var& query, const Var& keys, const Var& values, const Ten& maxLength, double dropout, bool trainDropout,
                                const Var& wQuery, const Var& wKeys, const Var& wValues, const Var& wOutput, int64_t numHeads,
                                bool linearized, bool causalMask);
};

struct TransformerEncoderBlock : Module {   // Transformer.scala:212-260; state: attention, layerNorm1, layerNorm2, w1, w2, b1, b2, scale1, scale2
  std::shared_ptr<MultiheadAttention> attention;
  std::shared_ptr<LayerNorm> layerNorm1, layerNorm2;
  Var w1, b1, w2, b2, scale1, scale2;
  double dropout = 0; bool train = true, gptOrder = false;
  static std::shared_ptr<TransformerEncoderBlock> make(int64_t in, int64_t attentionHiddenPerHeadDim, int64_t attentionNumHeads,
                                                       int64_t mlpHiddenDim, int64_t out, double dropout, int dtype, int device,
                                                       bool linearized, bool gptOrder, bool causalMask);
  void collect_state(std::vector<Var>& o) override;
  Var block(const Var& input, const Ten& maxLength);
  Var forward(const Var& x) override { return block(x, Ten()); }
  Var forward_multi(const std::vector<Var>& xs, const std::vector<Ten>& aux) override { return block(xs.at(0), aux.empty() ? Ten() : aux[0]); }
  void set_training(bool t) override { train = t; }
};

struct TransformerEncoder : Module {        // Transformer.scala:29-37
  std::vector<std::shared_ptr<TransformerEncoderBlock>> blocks;
  static std::shared_ptr<TransformerEncoder> make(int64_t numBlocks, int64_t in, int64_t attentionHiddenPerHeadDim, int64_t attentionNumHeads,
                                                  int64_t mlpHiddenDim, double dropout, int dtype, int device, bool linearized, bool gptOrder,
                                                  bool causalMask);
  void collect_state(std::vector<Var>& o) override { for (auto& b : blocks) b->collect_state(o); }
  Var encode(const Var& input, const Ten& maxLength) {
    Var a = input;
    for (auto& b : blocks) a = b->block(a, maxLength);
    return a;
  }
  Var forward(const Var& x) override { return encode(x, Ten()); }
  Var forward_multi(const std::vector<Var>& xs, const std::vector<Ten>& aux) override { return encode(xs.at(0), aux.empty() ? Ten() : aux[0]); }
  void set_training(bool t) override { for (auto& b : blocks) b->set_training(t); }
};

struct TransformerDecoderBlock : Module {   // Transformer.scala:263-307; state: attDD, attED, layerNorm1..4, w1, w2, b1, b2
  std::shared_ptr<MultiheadAttention> attentionDecoderDecoder, attentionEncoderDecoder;
  std::shared_ptr<LayerNorm> layerNorm1, layerNorm2, layerNorm3, layerNorm4;
  Var w1, b1, w2, b2;
  double dropout = 0; bool train = true;
  static std::shared_ptr<TransformerDecoderBlock> make(int64_t in, int64_t attentionHiddenPerHeadDim, int64_t attentionNumHeads,
                                                       int64_t mlpHiddenDim, int64_t out, double dropout, int dtype, int device,
                                                       bool linearized, bool decoderDecoderCausalMask, bool encoderDecoderCausalMask);
  void collect_state(std::vector<Var>& o) override;
  Var block(const Var& decoderInput, const Var& encoderOutput, const Ten& maxLength);
  Var forward(const Var&) override { LAMP_CHECK(false, "TransformerDecoderBlock takes (decoderInput, encoderOutput, maxLength?)"); return nullptr; }
  Var forward_multi(const std::vector<Var>& xs, const std::vector<Ten>& aux) override {
    LAMP_CHECK(xs.size() == 2, "TransformerDecoderBlock takes (decoderInput, encoderOutput, maxLength?)");
    return block(xs[0], xs[1], aux.empty() ? Ten() : aux[0]);
  }
  void set_training(bool t) override { train = t; }
};

struct TransformerDecoder : Module {        // Transformer.scala:105-114
  std::vector<std::shared_ptr<TransformerDecoderBlock>> blocks;
  void collect_state(std::vector<Var>& o) override { for (auto& b : blocks) b->collect_state(o); }
  Var decode(const Var& input, const Var& encoderOutput, const Ten& maxLength) {
    Var a = input;
    for (auto& b : blocks) a = b->block(a, encoderOutput, maxLength);
    return a;
  }
  Var forward(const Var&) override { LAMP_CHECK(false, "TransformerDecoder takes (decoderInput, encoderOutput, maxLength?)"); return nullptr; }
  Var forward_multi(const std::vector<Var>& xs, const std::vector<Ten>& aux) override {
    LAMP_CHECK(xs.size() == 2, "TransformerDecoder takes (decoderInput, encoderOutput, maxLength?)");
    return decode(xs[0], xs[1], aux.empty() ? Ten() : aux[0]);
  }
  void set_training(bool t) override { for (auto& b : blocks) b->set_training(t); }
};

struct Transformer : Module {               // Transformer.scala:310-379 (encoder in gpt order, decoder-decoder attention causal by default)
  std::shared_ptr<TransformerEncoder> encoder;
  std::shared_ptr<TransformerDecoder> decoder;
  static std::shared_ptr<Transformer> make(int64_t numBlocks, int64_t in, int64_t attentionHiddenPerHeadDim, int64_t attentionNumHeads,
                                           int64_t mlpHiddenDim, double dropout, int dtype, int device, bool linearized,
                                           bool encoderCausalMask, bool decoderDecoderCausalMask, bool encoderDecoderCausalMask);
  void collect_state(std::vector<Var>& o) override { encoder->collect_state(o); decoder->collect_state(o); }
  Var forward(const Var&) override { LAMP_CHECK(false, "Transformer takes (decoderInput, encoderInput, decoderMaxLength?, encoderMaxLength?)"); return nullptr; }
  // vars = (decoderInput, encoderInput); aux = (decoderMaxLength?, encoderMaxLength?)
  Var forward_multi(const std::vector<Var>& xs, const std::vector<Ten>& aux) override {
    LAMP_CHECK(xs.size() == 2, "Transformer takes (decoderInput, encoderInput, decoderMaxLength?, encoderMaxLength?)");
    Var encoderOutput = encoder->encode(xs[1], aux.size() > 1 ? aux[1] : Ten());
    return decoder->decode(xs[0], encoderOutput, aux.empty() ? Ten() : aux[0]);
  }
  void set_training(bool t) override { encoder->set_training(t); decoder->set_training(t); }
};

// PositionalEmbedding.vaswani (Transformer.scala:1022-1045): [sequenceLength, dimension], computed in f64 on the host
Ten positional_embedding_vaswani(int64_t sequenceLength, int64_t dimension, int dtype, int device);

struct TransformerEmbedding : Module {      // Transformer.scala:1105-1125; state: positionalEmbedding, embedding.weights
  std::shared_ptr<Embedding> embedding;
  bool addPositionalEmbedding = true;
  Var positionalEmbedding;
  void collect_state(std::vector<Var>& o) override { o.push_back(positionalEmbedding); embedding->collect_state(o); }
  Var forward(const Var& x) override;
};

struct LanguageModelModule : Module {       // lm.scala:137-190; state: tokenEmbedding, positionEmbedding, encoder, finalNorm
  std::shared_ptr<Embedding> tokenEmbedding, positionEmbedding;
  std::shared_ptr<TransformerEncoder> encoder;
  std::shared_ptr<LayerNorm> finalNorm;
  static std::shared_ptr<LanguageModelModule> make(int64_t maxLength, int64_t vocabularySize, int64_t numBlocks, int64_t embeddingDim,
                                                   int64_t attentionHiddenPerHeadDim, int64_t attentionNumHeads, int64_t encoderMlpHiddenDim,
                                                   double dropout, int dtype, int device, bool linearized);
  void collect_state(std::vector<Var>& o) override {
    tokenEmbedding->collect_state(o); positionEmbedding->collect_state(o); encoder->collect_state(o); finalNorm->collect_state(o);
  }
  // LanguageModelOutput(encoded, languageModelLogits)
  std::pair<Var, Var> run(const Var& tokens, const Ten& maxLength, const Ten& positions);
  Var forward(const Var& tokens) override { return run(tokens, Ten(), Ten()).second; }
  // vars = (tokens); aux = (maxLength?, positions?); returns the logits
  Var forward_multi(const std::vector<Var>& xs, const std::vector<Ten>& aux) override {
    return run(xs.at(0), aux.size() > 0 ? aux[0] : Ten(), aux.size() > 1 ? aux[1] : Ten()).second;
  }
  void set_training(bool t) override { encoder->set_training(t); }
};

struct LanguageModelLoss : Module {         // lm.scala:44-59: NLL(mean, ignore = padToken, class weights of ones) over logSoftMax(2).flatten(0, 1)
  std::shared_ptr<LanguageModelModule> languageModel;
  Ten classWeights; int64_t padToken = -100;
  static std::shared_ptr<LanguageModelLoss> make(int64_t maxLength, int64_t vocabularySize, int64_t numBlocks, int64_t embeddingDim,
                                                 int64_t attentionHiddenPerHeadDim, int64_t attentionNumHeads, int64_t encoderMlpHiddenDim,
                                                 double dropout, int64_t padToken, int dtype, int device, bool linearized);
  void collect_state(std::vector<Var>& o) override { languageModel->collect_state(o); }
  Var loss(const Var& tokens, const Ten& target, const Ten& maxLength, const Ten& positions);
  Var forward(const Var&) override { LAMP_CHECK(false, "LanguageModelLoss takes LossInput: (tokens), (languageModelTarget, maxLength?, positions?)"); return nullptr; }
  // vars = (tokens); aux = (languageModelTarget, maxLength?, positions?) - with LossFunctions.Identity the supervised model hands
  // its target over as aux[0] (train.scala of example-autoregressivelm: SupervisedModel(net, LossFunctions.Identity))
  Var forward_multi(const std::vector<Var>& xs, const std::vector<Ten>& aux) override {
    LAMP_CHECK(!aux.empty() && aux[0].defined(), "LanguageModelLoss needs the languageModelTarget");
    return loss(xs.at(0), aux[0], aux.size() > 1 ? aux[1] : Ten(), aux.size() > 2 ? aux[2] : Ten());
  }
  void set_training(bool t) override { languageModel->set_training(t); }
};

}  // namespace host
}  // namespace lamp
