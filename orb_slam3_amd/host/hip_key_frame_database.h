// GPU-backed query for frame::DBoW2Database (src/frame/database/DBoW2/dbo_w2_database.{h,cpp}).  The database keeps its
// inverted_file_ and Serialize exactly as they are; this index is the device-resident, keyframe-major copy of the same
// data that answers SearchWordSharingKeyFrames (:83-103) and the vocabulary->score calls of
// FilterCandidatesByWordSharingAcceptableScore (:114-132) for ALL keyframes in one C-ABI call.  INTEGRATION.md shows
// the three places DBoW2Database calls it from (Append, Erase, DetectNBestCandidates).
//
// BowVector is DBoW2::BowVector (std::map<WordId, WordValue>), WordFrequencies the handler's
// std::map<unsigned, std::size_t> (dbo_w2_handler.h:33); any ordered map with those value types works.
#ifndef ORB_SLAM3_AMD_HIP_KEY_FRAME_DATABASE_H_
#define ORB_SLAM3_AMD_HIP_KEY_FRAME_DATABASE_H_

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "orbgpu.h"

namespace orb_slam3 {
namespace frame {

class HipKeyFrameIndex {
 public:
  struct Hit {
    uint64_t keyframe_id;   // KeyFrame::Id()
    uint64_t shared_words;  // WordSharingKeyFrameMap value; 0 = not in the map
    double score;           // vocabulary->score(query BowVector, keyframe BowVector)
  };

  // scoring = DBoW2::ScoringType of the vocabulary (L1_NORM for ORBvoc); with_counts only if word frequencies other
  // than 1 can occur (DBoW2Handler::Precompute never produces them)
  HipKeyFrameIndex(orbgpu_ctx * ctx, int scoring, bool with_counts = false) : ctx_(ctx), with_counts_(with_counts) {
    if (orbgpu_bowdb_create(ctx, scoring, with_counts ? 1 : 0, &db_) != ORBGPU_OK)
      throw std::runtime_error(std::string("orbgpu_bowdb_create: ") + orbgpu_last_error(ctx));
  }
  ~HipKeyFrameIndex() { orbgpu_bowdb_destroy(db_); }
  HipKeyFrameIndex(const HipKeyFrameIndex &) = delete;
  HipKeyFrameIndex & operator=(const HipKeyFrameIndex &) = delete;

  // DBoW2Database::Append, :53-61
  template <class BowVector, class WordFrequencies>
  void Append(uint64_t keyframe_id, const BowVector & bow, const WordFrequencies * word_frequencies) {
    Flatten(bow, word_frequencies);
    if (orbgpu_bowdb_append(db_, keyframe_id, words_.data(), values_.data(), with_counts_ ? counts_.data() : nullptr,
                            (int32_t)words_.size()) != ORBGPU_OK)
      throw std::runtime_error(std::string("orbgpu_bowdb_append: ") + orbgpu_last_error(ctx_));
  }
  template <class BowVector>
  void Append(uint64_t keyframe_id, const BowVector & bow) {
    Append<BowVector, BowVector>(keyframe_id, bow, nullptr);
  }

  // DBoW2Database::Erase, :187-195
  void Erase(uint64_t keyframe_id) {
    if (orbgpu_bowdb_erase(db_, keyframe_id) != ORBGPU_OK)
      throw std::runtime_error(std::string("orbgpu_bowdb_erase: ") + orbgpu_last_error(ctx_));
  }

  size_t Size() const {
    int32_t n = 0;
    orbgpu_bowdb_size(db_, &n);
    return (size_t)n;
  }

  // one entry per keyframe of the database, in no particular order
  template <class BowVector, class WordFrequencies>
  std::vector<Hit> Query(const BowVector & bow, const WordFrequencies * word_frequencies) {
    Flatten(bow, word_frequencies);
    const size_t n = Size();
    std::vector<uint64_t> ids(n), shared(n);
    std::vector<double> score(n);
    int32_t n_out = 0;
    if (orbgpu_bowdb_query(db_, words_.data(), values_.data(), word_frequencies ? counts_.data() : nullptr,
                           (int32_t)words_.size(), (int32_t)n, ids.data(), shared.data(), score.data(), &n_out) != ORBGPU_OK)
      throw std::runtime_error(std::string("orbgpu_bowdb_query: ") + orbgpu_last_error(ctx_));
    std::vector<Hit> hits((size_t)n_out);
    for (size_t k = 0; k < hits.size(); ++k) hits[k] = Hit{ids[k], shared[k], score[k]};
    return hits;
  }
  template <class BowVector>
  std::vector<Hit> Query(const BowVector & bow) {
    return Query<BowVector, BowVector>(bow, nullptr);
  }

 private:
  // std::map order = ascending word id, which is what the C ABI asks for; a word without a frequency entry counts 0
  template <class BowVector, class WordFrequencies>
  void Flatten(const BowVector & bow, const WordFrequencies * wf) {
    words_.clear();
    values_.clear();
    counts_.clear();
    for (const auto & kv : bow) {
      words_.push_back((uint32_t)kv.first);
      values_.push_back((double)kv.second);
      uint32_t c = 1;
      if (wf) {
        auto it = wf->find(kv.first);
        c = it == wf->end() ? 0u : (uint32_t)it->second;
      }
      counts_.push_back(c);
    }
  }

  orbgpu_ctx * ctx_;
  orbgpu_bowdb * db_ = nullptr;
  bool with_counts_;
  std::vector<uint32_t> words_, counts_;
  std::vector<double> values_;
};

}  // namespace frame
}  // namespace orb_slam3
#endif
