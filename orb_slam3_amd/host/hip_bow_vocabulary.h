// GPU-backed body for DBoW2Handler::Precompute (src/features/handlers/DBoW2/dbo_w2_handler.cpp:74-93), i.e.
// TemplatedVocabulary<FORB::TDescriptor, FORB>::transform(features, BowVector, FeatureVector, levelsup)
// (DBoW2/DBoW2/TemplatedVocabulary.h:1125-1189).  The per-feature tree descent (:1218-1259, FORB::distance
// FORB.cpp:81-101) is one batched C-ABI call; the BowVector / FeatureVector bookkeeping is the reference's own
// (addWeight / addIfNotExist / addFeature / normalize are called on the reference's types).
//
// The vocabulary keeps its tree private (m_nodes); a maintainer adds ONE accessor inside TemplatedVocabulary that
// flattens it (shown in INTEGRATION.md) and hands the arrays to this class once, after load.
#ifndef ORB_SLAM3_AMD_HIP_BOW_VOCABULARY_H_
#define ORB_SLAM3_AMD_HIP_BOW_VOCABULARY_H_

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "orbgpu.h"

namespace orb_slam3 {
namespace features {
namespace handlers {

enum class HipBowWeighting { TF_IDF, TF, IDF, BINARY };  // DBoW2::WeightingType
enum class HipBowNorm { NONE, L1, L2 };                  // GeneralScoring::mustNormalize(norm)

class HipBowVocabulary {
 public:
  // child_off / children / node_desc / word_id as documented at orbgpu_vocab_create; weight[v] = m_nodes[v].weight
  HipBowVocabulary(orbgpu_ctx * ctx, const std::vector<int32_t> & child_off, const std::vector<uint32_t> & children,
                   const std::vector<uint8_t> & node_desc, const std::vector<uint32_t> & word_id,
                   const std::vector<double> & weight, int levels, HipBowWeighting weighting, HipBowNorm norm)
      : ctx_(ctx), weighting_(weighting), norm_(norm) {
    const int n_nodes = (int)word_id.size();
    if (orbgpu_vocab_create(ctx, n_nodes, child_off.data(), children.data(), node_desc.data(), word_id.data(), levels,
                            &vocab_) != ORBGPU_OK)
      throw std::runtime_error(std::string("orbgpu_vocab_create: ") + orbgpu_last_error(ctx));
    for (int v = 0; v < n_nodes; ++v)
      if (child_off[v + 1] == child_off[v]) {
        if (word_weight_.size() <= word_id[v]) word_weight_.resize(word_id[v] + 1, 0.0);
        word_weight_[word_id[v]] = weight[v];
      }
  }
  ~HipBowVocabulary() { orbgpu_vocab_destroy(vocab_); }
  HipBowVocabulary(const HipBowVocabulary &) = delete;
  HipBowVocabulary & operator=(const HipBowVocabulary &) = delete;

  // descriptors: n x 32 bytes, row-major (Features::descriptors.data()).  BowVector / FeatureVector are
  // DBoW2::BowVector / DBoW2::FeatureVector.
  template <class BowVector, class FeatureVector>
  void transform(const uint8_t * descriptors, size_t n, BowVector & v, FeatureVector & fv, int levelsup) const {
    v.clear();
    fv.clear();
    if (n == 0) return;
    std::vector<uint32_t> word(n), node(n);
    if (orbgpu_bow_transform(ctx_, vocab_, descriptors, (int32_t)n, levelsup, word.data(), node.data()) != ORBGPU_OK)
      throw std::runtime_error(std::string("orbgpu_bow_transform: ") + orbgpu_last_error(ctx_));
    const bool tf_like = weighting_ == HipBowWeighting::TF_IDF || weighting_ == HipBowWeighting::TF;
    for (size_t i = 0; i < n; ++i) {
      const double w = word_weight_[word[i]];
      if (w > 0) {  // not stopped, TemplatedVocabulary.h:1152,1180
        if (tf_like) v.addWeight(word[i], w); else v.addIfNotExist(word[i], w);
        fv.addFeature(node[i], (unsigned)i);
      }
    }
    if (tf_like && !v.empty() && norm_ == HipBowNorm::NONE) {  // :1159-1165
      const double nd = (double)v.size();
      for (auto & kv : v) kv.second /= nd;
    }
    if (norm_ != HipBowNorm::NONE) v.normalize(norm_ == HipBowNorm::L1 ? BowVector::kL1 : BowVector::kL2);  // :1189
  }

 private:
  orbgpu_ctx * ctx_;
  orbgpu_vocab * vocab_ = nullptr;
  HipBowWeighting weighting_;
  HipBowNorm norm_;
  std::vector<double> word_weight_;
};

}  // namespace handlers
}  // namespace features
}  // namespace orb_slam3
#endif
