// Drives the C++ adapters (orb_slam3_amd/host/) the way the reference's callers do and dumps the results for
// tests/test_gpu_host_adapter.py to compare with the oracle.  usage: adapter_test W H NFEAT img0.raw img1.raw out.bin
#include <cmath>
#include <cstdio>
#include <fstream>
#include <sstream>
#include <list>
#include <map>
#include <memory>

#include "hip_orb_feature_extractor.h"
// the same-name replacements of the reference's matcher headers (orb_slam3_amd/host/features/...): included by the
// paths the reference's callers use
#include <features/matching/second_nearest_neighbor_matcher.hpp>
#include <features/matching/iterators/area_to_iterator.h>
#include <features/matching/iterators/projection_search_iterator.h>
#include <features/matching/validators/orientation_validator.h>
#include <features/handlers/DBoW2/bow_to_iterator.h>
#include "hip_snn_matcher.h"
#include "hip_bow_vocabulary.h"
#include "hip_map_point.h"
#include "hip_monocular_camera.h"
#include "hip_key_frame_database.h"

using namespace orb_slam3;

// stand-ins for DBoW2::BowVector / DBoW2::FeatureVector (BowVector.cpp:34-84, FeatureVector.cpp:33-47): the
// adapter only needs their public operations
struct TestBowVector : std::map<unsigned, double> {
  enum { kL1, kL2 };
  void addWeight(unsigned id, double v) { auto it = lower_bound(id); if (it != end() && it->first == id) it->second += v; else insert(it, {id, v}); }
  void addIfNotExist(unsigned id, double v) { auto it = lower_bound(id); if (it == end() || it->first != id) insert(it, {id, v}); }
  void normalize(int type) {
    double norm = 0.0;
    if (type == kL1) { for (auto & kv : *this) norm += std::fabs(kv.second); }
    else { for (auto & kv : *this) norm += kv.second * kv.second; norm = std::sqrt(norm); }
    if (norm > 0.0) for (auto & kv : *this) kv.second /= norm;
  }
};
struct TestFeatureVector : std::map<unsigned, std::vector<unsigned>> {
  void addFeature(unsigned id, unsigned i) { (*this)[id].push_back(i); }
};

static TImageGray8U load(const char * path, int w, int h) {
  TImageGray8U img;
  img.resize(h, w);
  std::ifstream f(path, std::ios::binary);
  f.read((char *)img.data(), (std::streamsize)w * h);
  return img;
}

template <class T>
static void put(std::ofstream & o, const T & v) { o.write((const char *)&v, sizeof(T)); }

int main(int argc, char ** argv) {
  if (argc != 7) return 2;
  const int w = atoi(argv[1]), h = atoi(argv[2]), nfeat = atoi(argv[3]);
  // the driver's construction, main.cpp:451-454, through the interface pointer Atlas owns (map/atlas.h:44)
  std::unique_ptr<features::IFeatureExtractor> fe(new features::HipORBFeatureExtractor(w, h, 1.2, 8, 20, 7, 0, nfeat));
  auto * hip = static_cast<features::HipORBFeatureExtractor *>(fe.get());
  features::Features f0(w, h), f1(w, h);
  TImageGray8U i0 = load(argv[4], w, h), i1 = load(argv[5], w, h), empty;
  if (fe->Extract(empty, f0, nfeat) != -1) return 3;  // orb_feature_extractor.cpp:533
  if (fe->Extract(i0, f0, nfeat) != 0 || fe->Extract(i1, f1, nfeat) != 0) return 4;
  // MonocularFrame::ComputeMatchesForLinking, monocular_frame.cpp:173-189, statement for statement (feature_handler_->
  // GetFeatures() is f1, from_frame's is f0, GetAtlas()->GetFeatureExtractor() is fe.get())
  std::unordered_map<std::size_t, std::size_t> matches;
  size_t n_before;
  int discarded;
  {
    std::unordered_map<size_t, size_t> & out_matches = matches;
    features::matching::SNNMatcher<features::matching::iterators::AreaToIterator> matcher(0.9, 50);
    features::matching::iterators::AreaToIterator
        begin(0, &f1, &f0, 100);
    features::matching::iterators::AreaToIterator
        end(f1.Size(),
            &f1,
            &f0,
            100);

    matcher.MatchWithIterators(begin, end, fe.get(), out_matches);
    n_before = out_matches.size();
    discarded = features::matching::OrientationValidator
        (f1.keypoints,
         f0.keypoints).Validate(
        out_matches);
  }
  {  // the same through the workhorse class must agree
    std::unordered_map<std::size_t, std::size_t> m2;
    features::matching::HipSNNMatcher(hip->context(), 0.9, 50).MatchArea(f1, f0, 100, m2);
    features::matching::HipOrientationValidator(hip->context(), f1.keypoints, f0.keypoints).Validate(m2);
    if (m2 != matches) return 5;
  }
  // SearchInVisiblePoints, monocular_frame.cpp:285-306: one "map point" per level-0/1 keypoint of frame 0
  std::vector<map::MapPoint> mps(f0.Size());
  std::list<frame::MapPointVisibilityParams> vis;
  for (size_t j = 0; j < f0.Size(); ++j) {
    if (f0.keypoints[j].level > 1) continue;
    std::memcpy(mps[j].desc.d, f0.descriptors.data() + 32 * j, 32);
    frame::MapPointVisibilityParams v{};
    v.map_point = &mps[j];
    v.window_size = 15.7 * fe->GetScaleFactors()[(size_t)f0.keypoints[j].level];
    v.position = f0.keypoints[j].pt;
    v.level = f0.keypoints[j].level;
    vis.push_back(v);
  }
  // MonocularFrame::SearchInVisiblePoints, monocular_frame.cpp:285-306, statement for statement
  std::unordered_map<map::MapPoint *, std::size_t> proj;
  {
    const std::list<frame::MapPointVisibilityParams> & filtered_map_points = vis;
    const double matcher_snn_threshold = 0.9;
    features::matching::iterators::ProjectionSearchIterator begin
        (filtered_map_points.begin(),
         filtered_map_points.end(),
         &f1);

    features::matching::iterators::ProjectionSearchIterator end
        (filtered_map_points.end(),
         filtered_map_points.end(),
         &f1);

    typedef features::matching::SNNMatcher<features::matching::iterators::ProjectionSearchIterator> Matcher;
    Matcher matcher(matcher_snn_threshold, 100 /* constants::MONO_TWMM_THRESHOLD_HIGH */);
    Matcher::MatchMapType matches;
    matcher.MatchWithIterators(begin, end, fe.get(), matches);
    proj = matches;
  }

  // DBoW2Handler::FastMatch(WEAK), dbo_w2_handler.cpp:25-72, with synthetic feature vectors: node = first
  // descriptor byte mod 16 (the vocabulary itself is outside the path)
  DBoW2::FeatureVector fv0, fv1;
  for (size_t j = 0; j < f0.Size(); ++j) fv0.addFeature(f0.descriptors.data()[32 * j] % 16u, (unsigned)j);
  for (size_t j = 0; j < f1.Size(); ++j) fv1.addFeature(f1.descriptors.data()[32 * j] % 16u, (unsigned)j);
  std::unordered_map<std::size_t, std::size_t> bow;
  {  // dbo_w2_handler.cpp:29-71 with severity WEAK and check_orientation = false
    typedef features::matching::SNNMatcher<features::handlers::iterators::BowToIterator> BOW_MATCHER;
    std::unordered_map<std::size_t, std::size_t> & out_matches = bow;
    const DBoW2::FeatureVector & feature_vector_ = fv1;
    const unsigned threshold = (unsigned)fe->GetHighThreshold();
    const double ratio = 0.9;
    BOW_MATCHER bow_matcher(ratio, threshold);
    features::handlers::iterators::BowToIterator bow_it_begin(feature_vector_.begin(),
                                                              &feature_vector_,
                                                              &fv0,
                                                              &f1,
                                                              &f0);

    features::handlers::iterators::BowToIterator bow_it_end(feature_vector_.end(),
                                                            &feature_vector_,
                                                            &fv0,
                                                            &f1,
                                                            &f0);

    bow_matcher.MatchWithIterators(bow_it_begin, bow_it_end, fe.get(), out_matches);
  }

  // DBoW2Handler::Precompute, dbo_w2_handler.cpp:74-93, on a small regular vocabulary (k = 4, L = 3) whose node
  // descriptors are rows of frame 0's descriptor matrix (the test rebuilds the same tree)
  std::vector<int32_t> child_off{0};
  std::vector<uint32_t> children, word_id;
  std::vector<uint8_t> node_desc;
  std::vector<double> weight;
  {
    const int k = 4, L = 3;
    int n_nodes = 0, width = 1;
    for (int l = 0; l <= L; ++l) { n_nodes += width; width *= k; }
    int next = 1, first_leaf = n_nodes - width / k;
    for (int v = 0; v < n_nodes; ++v) {
      if (v < first_leaf) for (int c = 0; c < k; ++c) children.push_back((uint32_t)next++);
      child_off.push_back((int32_t)children.size());
      const uint8_t * src = f0.descriptors.data() + 32 * (size_t)(v % f0.Size());
      node_desc.insert(node_desc.end(), src, src + 32);
      word_id.push_back(v >= first_leaf ? (uint32_t)(v - first_leaf) : 0u);
      weight.push_back(v % 7 == 3 ? 0.0 : 0.5 + 0.25 * (v % 5));
    }
  }
  features::handlers::HipBowVocabulary voc(hip->context(), child_off, children, node_desc, word_id, weight, 3,
                                           features::handlers::HipBowWeighting::TF_IDF, features::handlers::HipBowNorm::L1);
  TestBowVector bv;
  TestFeatureVector tfv;
  voc.transform(f1.descriptors.data(), f1.Size(), bv, tfv, 2);

  // MapPoint::ComputeDistinctiveDescriptor, map_point.cpp:135-171: "map point" g is observed by the descriptors
  // g, g+7, g+14, ... of frame 0 (g < 7; sizes differ by one, some even, some odd)
  std::vector<std::vector<features::DescriptorType>> groups(8);   // the last group stays empty
  for (size_t j = 0; j < f0.Size(); ++j) {
    features::DescriptorType d;
    std::memcpy(d.d, f0.descriptors.data() + 32 * j, 32);
    groups[j % 7].push_back(d);
  }
  const std::vector<int32_t> distinctive = map::HipComputeDistinctiveDescriptors(hip->context(), groups);

  // FeatureHandlerFactory::Create, feature_handler_factory.cpp:53-60, with a Barrel5 camera
  orbgpu_camera cam{};
  cam.fx = 258.6; cam.fy = 258.2; cam.cx = 160.2; cam.cy = 120.1; cam.model = ORBGPU_DISTORTION_BARREL5;
  cam.k[0] = -0.28; cam.k[1] = 0.07; cam.k[2] = 2e-4; cam.k[3] = 1.8e-5; cam.k[4] = 0.0;
  const std::vector<uint8_t> geom_ok = camera::HipUndistortFeatures(hip->context(), cam, f1);

  // DBoW2Database: keyframe g holds the words (j * 13) % 97 of the descriptors j = g, g + 9, ... of frame 0, value =
  // L1-normalised byte 1 of the descriptor; keyframe 4 is erased again, keyframe 2 re-appended; query = keyframe 6's vector
  frame::HipKeyFrameIndex index(hip->context(), 0 /* L1_NORM */);
  std::vector<std::map<unsigned, double>> kf_bow(9);
  for (size_t g = 0; g < kf_bow.size(); ++g) {
    for (size_t j = g; j < f0.Size(); j += 9) kf_bow[g][(unsigned)((j * 13) % 97)] += 1.0 + f0.descriptors.data()[32 * j + 1];
    double sum = 0;
    for (const auto & kv : kf_bow[g]) sum += std::fabs(kv.second);
    for (auto & kv : kf_bow[g]) kv.second /= sum;
    index.Append(100 + g, kf_bow[g]);
  }
  index.Erase(104);
  index.Erase(555);  // absent: no-op
  index.Append(102, kf_bow[2]);
  const std::vector<frame::HipKeyFrameIndex::Hit> hits = index.Query(kf_bow[6]);

  std::ofstream o(argv[6], std::ios::binary);
  for (const features::Features * f : {&f0, &f1}) {
    put(o, (int32_t)f->Size());
    for (const auto & k : f->keypoints) { put(o, k.pt[0]); put(o, k.pt[1]); put(o, k.angle); put(o, k.size); put(o, (int32_t)k.level); }
    o.write((const char *)f->descriptors.data(), (std::streamsize)f->Size() * 32);
  }
  put(o, (int32_t)n_before); put(o, (int32_t)discarded); put(o, (int32_t)matches.size());
  for (const auto & kv : matches) { put(o, (int32_t)kv.first); put(o, (int32_t)kv.second); }
  put(o, (int32_t)proj.size());
  for (const auto & kv : proj) { put(o, (int32_t)(kv.first - mps.data())); put(o, (int32_t)kv.second); }
  put(o, (int32_t)bow.size());
  for (const auto & kv : bow) { put(o, (int32_t)kv.first); put(o, (int32_t)kv.second); }
  put(o, (int32_t)bv.size());
  for (const auto & kv : bv) { put(o, (int32_t)kv.first); put(o, kv.second); }
  put(o, (int32_t)tfv.size());
  for (const auto & kv : tfv) { put(o, (int32_t)kv.first); put(o, (int32_t)kv.second.size()); for (unsigned i : kv.second) put(o, (int32_t)i); }
  put(o, (int32_t)distinctive.size());
  for (int32_t b : distinctive) put(o, b);
  put(o, (int32_t)geom_ok.size());
  for (size_t i = 0; i < geom_ok.size(); ++i) {
    put(o, geom_ok[i]);
    put(o, f1.undistorted_keypoints[i][0]); put(o, f1.undistorted_keypoints[i][1]);
    for (int c = 0; c < 3; ++c) put(o, f1.undistorted_and_unprojected_keypoints[i][c]);
  }
  put(o, (int32_t)hits.size());
  for (const auto & h : hits) { put(o, h.keyframe_id); put(o, h.shared_words); put(o, h.score); }
  // scalar helpers
  put(o, (uint32_t)fe->PredictScale(1.0, 3.0)); put(o, fe->GetAcceptableSquareError(3));
  features::DescriptorType a, b; std::memcpy(a.d, f0.descriptors.data(), 32); std::memcpy(b.d, f0.descriptors.data() + 32, 32);
  put(o, (uint32_t)fe->ComputeDistance(a, b));
  std::ostringstream ss; fe->Serialize(ss); put(o, (int32_t)ss.str().size());
  return 0;
}
