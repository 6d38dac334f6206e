// Keyframe-database query: shared-word counts and BowVector scores of one query against every keyframe of the
// database in one launch.  gfx950 only.
//
// Reference: DBoW2Database::SearchWordSharingKeyFrames (src/frame/database/DBoW2/dbo_w2_database.cpp:83-103) walks the
// inverted file word by word and adds kf_count * query_count per (word, keyframe); FilterCandidatesByWordSharing-
// AcceptableScore (:114-132) then calls vocabulary->score(query BowVector, keyframe BowVector) per surviving keyframe
// = L1Scoring / L2Scoring / ChiSquareScoring / BhattacharyyaScoring / DotProductScoring::score
// (DBoW2/DBoW2/ScoringObject.cpp:23-68, :73-120, :125-170, :226-266, :271-311): a merge over the two ascending word
// lists that accumulates one double per common word IN WORD ORDER.  (KLScoring, :175-221, also sums over the words
// only the query holds and goes through log(); it is not offered.)
// Here the database is held keyframe-major (the transpose of the inverted file): one wave per keyframe, 64 of its
// words per step, each lane looks its word up in the query (a 128 Kbit membership filter in LDS answers "not in the
// query" with one read for almost every word; the rest take a binary search in LDS), and the per-word terms are added
// to the running double strictly in word order (ballot + readlane), so the sum is the reference's bit for bit.
#include "orbgpu_internal.h"

namespace {

constexpr int BS_WAVES = 4;
constexpr int BS_FILTER_DW = 4096;  // membership filter: bit (word id mod 2^17) -- word ids are dense small integers

__device__ __forceinline__ double readlane_f64(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

template <bool Q_LDS>
__global__ __launch_bounds__(64 * BS_WAVES) void k_bow_score(BowScoreDev a, unsigned long long *__restrict__ shared_out,
                                                             double *__restrict__ score_out) {
  extern __shared__ unsigned s_dyn[];  // Q_LDS: membership filter, then the query's word ids
  unsigned *s_filter = s_dyn, *s_qw = s_dyn + BS_FILTER_DW;
  if (Q_LDS) {
    for (int i = threadIdx.x; i < BS_FILTER_DW; i += 64 * BS_WAVES) s_filter[i] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < a.n_q; i += 64 * BS_WAVES) {
      const unsigned w = a.q_words[i];
      s_qw[i] = w;
      atomicOr(&s_filter[(w >> 5) & (BS_FILTER_DW - 1)], 1u << (w & 31));
    }
    __syncthreads();
  }
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int k = blockIdx.x * BS_WAVES + wave;
  if (k >= a.n_keyframes) return;
  const long long beg = a.kf_beg[k], end = a.kf_end[k];
  double acc = 0.0;
  unsigned long long sh = 0;
  constexpr int U = 8;  // 64-word steps whose loads are in flight together (the loop is latency-bound otherwise)
  for (long long base0 = beg; base0 < end; base0 += 64 * U) {
    unsigned wv[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const long long i = base0 + 64 * u + lane;
      wv[u] = i < end ? a.kf_words[i] : 0u;
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const long long i = base0 + 64 * u + lane;
      if (base0 + 64 * u >= end) break;  // wave-uniform
      bool hit = false;
      double term = 0.0;
      if (i < end) {
        const unsigned w = wv[u];
        // the filter has no false negatives: a clear bit means the word is not in the query
        const bool maybe = !Q_LDS || ((s_filter[(w >> 5) & (BS_FILTER_DW - 1)] >> (w & 31)) & 1u);
        int lo = 0, hi = maybe ? a.n_q : 0;  // std::map::lower_bound on the query
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          const unsigned qw = Q_LDS ? s_qw[mid] : a.q_words[mid];
          if (qw < w) lo = mid + 1; else hi = mid;
        }
        if (maybe && lo < a.n_q && (Q_LDS ? s_qw[lo] : a.q_words[lo]) == w) {
          hit = true;
          const double vi = a.q_values[lo], wi = a.kf_values[i];  // score(v1 = query, v2 = keyframe)
          switch (a.scoring) {
            case 0: term = (fabs(vi - wi) - fabs(vi)) - fabs(wi); break;       // L1_NORM, :41
            case 2: hit = vi + wi != 0.0; term = vi * wi / (vi + wi); break;   // CHI_SQUARE, :148
            case 4: term = sqrt(vi * wi); break;                               // BHATTACHARYYA, :245
            default: term = vi * wi; break;                                    // L2_NORM :91, DOT_PRODUCT :290
          }
          const unsigned long long qc = a.q_counts ? a.q_counts[lo] : 1u, kc = a.kf_counts ? a.kf_counts[i] : 1u;
          sh += kc * qc;
        }
      }
      unsigned long long m = __ballot(hit);
      while (m) {  // wave-uniform: common words in ascending order
        const int b = __builtin_ctzll(m);
        m &= m - 1;
        acc += readlane_f64(term, b);
      }
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const unsigned lo = __shfl_xor((unsigned)sh, d), hi = __shfl_xor((unsigned)(sh >> 32), d);
    sh += ((unsigned long long)hi << 32) | lo;
  }
  if (lane == 0) {
    double s;
    if (a.scoring == 0) s = -acc / 2.0;                                  // ScoringObject.cpp:65
    else if (a.scoring == 1) s = acc >= 1 ? 1.0 : 1.0 - sqrt(1.0 - acc); // :114-117
    else if (a.scoring == 2) s = 2. * acc;                               // :167
    else s = acc;                                                        // :265, :310
    shared_out[k] = sh;
    score_out[k] = s;
  }
}

// a resident database moves to a fresh arena when erased keyframes have left too many holes: one workgroup per keyframe
// copies its slice (counts may be absent)
__global__ __launch_bounds__(256) void k_bowdb_compact(const long long *__restrict__ old_beg, const long long *__restrict__ new_beg,
                                                       const long long *__restrict__ new_end, const unsigned *__restrict__ w_in,
                                                       const double *__restrict__ v_in, const unsigned *__restrict__ c_in,
                                                       unsigned *__restrict__ w_out, double *__restrict__ v_out,
                                                       unsigned *__restrict__ c_out) {
  const int k = blockIdx.x;
  const long long src = old_beg[k], dst = new_beg[k], n = new_end[k] - dst;
  for (long long i = threadIdx.x; i < n; i += 256) {
    w_out[dst + i] = w_in[src + i];
    v_out[dst + i] = v_in[src + i];
    if (c_in) c_out[dst + i] = c_in[src + i];
  }
}

}  // namespace

size_t bow_score_lds_bytes(int n_q) { return (size_t)n_q * 4 <= 44 * 1024 ? (size_t)(BS_FILTER_DW + std::max(n_q, 1)) * 4 : 0; }

void launch_bow_score(const orbgpu_ctx *c, const BowScoreDev &a, unsigned long long *d_shared, double *d_score) {
  const int grid = (a.n_keyframes + BS_WAVES - 1) / BS_WAVES;
  const size_t lds = bow_score_lds_bytes(a.n_q);
  if (lds)
    k_bow_score<true><<<grid, 64 * BS_WAVES, lds, c->stream>>>(a, d_shared, d_score);
  else
    k_bow_score<false><<<grid, 64 * BS_WAVES, 0, c->stream>>>(a, d_shared, d_score);
}

void launch_bowdb_compact(const orbgpu_ctx *c, int n_keyframes, const long long *d_old_beg, const long long *d_new_beg,
                          const long long *d_new_end, const unsigned *w_in, const double *v_in, const unsigned *c_in,
                          unsigned *w_out, double *v_out, unsigned *c_out) {
  if (n_keyframes > 0)
    k_bowdb_compact<<<n_keyframes, 256, 0, c->stream>>>(d_old_beg, d_new_beg, d_new_end, w_in, v_in, c_in, w_out, v_out, c_out);
}
