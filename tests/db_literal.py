"""Literal, container-for-container Python transcription of the reference's keyframe-database query
(src/frame/database/DBoW2/dbo_w2_database.cpp, DBoW2/DBoW2/ScoringObject.cpp) and a random database generator.
Shared by the CPU oracle tests and the GPU parity tests; it is a checker, never a product path."""
import math

import numpy as np


def random_database(seed, n_kf, vocab, words_per_kf, weighted_counts=False, dense_overlap=False):
    """-> (bows {kf id: {word: value}}, freqs {kf id: {word: count}}): L1-normalised BowVectors as transform() leaves them."""
    rng = np.random.default_rng(seed)
    bows, freqs = {}, {}
    pool = rng.choice(vocab, size=min(vocab, 3 * words_per_kf), replace=False) if dense_overlap else None
    for k in range(n_kf):
        n = int(rng.integers(0, words_per_kf + 1)) if k % 7 == 3 else int(rng.integers(max(1, words_per_kf // 2), words_per_kf + 1))
        src = pool if dense_overlap else vocab
        words = np.sort(rng.choice(src, size=min(n, len(pool) if dense_overlap else vocab), replace=False)) if n else np.zeros(0, np.int64)
        vals = rng.uniform(0.05, 4.0, len(words))
        s = 0.0
        for v in vals:
            s += math.fabs(v)
        bow = {int(w): (float(v) / s if s > 0 else float(v)) for w, v in zip(words, vals)}
        kid = 10 + 3 * k          # ids are not 0..n-1 in a real atlas
        bows[kid] = bow
        freqs[kid] = {w: (int(rng.integers(1, 5)) if weighted_counts else 1) for w in bow}
    return bows, freqs


def l_score(v1, v2, scoring):
    """{L1,L2,ChiSquare,Bhattacharyya,DotProduct}Scoring::score with std::map iteration = sorted keys."""
    k1, k2 = sorted(v1), sorted(v2)
    i1 = i2 = 0
    score = 0.0
    import bisect
    while i1 != len(k1) and i2 != len(k2):
        vi, wi = v1[k1[i1]], v2[k2[i2]]
        if k1[i1] == k2[i2]:
            if scoring == 0:
                score += math.fabs(vi - wi) - math.fabs(vi) - math.fabs(wi)
            elif scoring in (1, 5):
                score += vi * wi
            elif scoring == 2:
                if vi + wi != 0.0:
                    score += vi * wi / (vi + wi)
            elif scoring == 4:
                score += math.sqrt(vi * wi)
            else:
                raise ValueError
            i1 += 1
            i2 += 1
        elif k1[i1] < k2[i2]:
            i1 = bisect.bisect_left(k1, k2[i2])
        else:
            i2 = bisect.bisect_left(k2, k1[i1])
    if scoring == 0:
        score = -score / 2.0
    elif scoring == 1:
        score = 1.0 if score >= 1 else 1.0 - math.sqrt(1.0 - score)
    elif scoring == 2:
        score = 2. * score
    return score


class LiteralDatabase:
    def __init__(self, vocab_size):
        self.inverted_file = [dict() for _ in range(vocab_size)]
        self.bow = {}

    def Append(self, kf, bow, wf):                       # :53-61
        for w, c in wf.items():
            self.inverted_file[w][kf] = c
        self.bow[kf] = bow

    def Erase(self, kf):                                 # :187-195
        for w in self.bow[kf]:
            self.inverted_file[w].pop(kf, None)
        del self.bow[kf]

    def SearchWordSharingKeyFrames(self, q_wf):          # :83-103
        out = {}
        for w, c in sorted(q_wf.items()):
            for kf, kc in self.inverted_file[w].items():
                out[kf] = (out.get(kf, 0) + kc * c) & ((1 << 64) - 1)
        return out

    def DetectNBestCandidates(self, q_bow, q_wf, scoring, covisible, map_of, current_map, count):   # :63-179
        ws = self.SearchWordSharingKeyFrames(q_wf)
        if not ws:
            return set(), set()
        max_sharing = max(ws.values())
        min_common = int(0.8 * max_sharing)
        reloc = {kf: l_score(q_bow, self.bow[kf], scoring) for kf in sorted(ws) if ws[kf] > min_common}
        acc, best_acc = {}, 0.0
        for kf in sorted(reloc):
            best_score = reloc[kf]
            acc_score = best_score
            best_kf = kf
            for nb in covisible(kf):
                if nb not in ws:
                    continue
                if nb in reloc:
                    acc_score += reloc[nb]
                    if reloc[nb] > best_score:
                        best_kf, best_score = nb, reloc[nb]
            acc[best_kf] = acc_score
            if acc_score > best_acc:
                best_acc = acc_score
        loop, merge = set(), set()
        for kf, sc in acc.items():
            if sc > 0.75 * best_acc:
                if map_of(kf) == current_map and len(loop) < count:
                    loop.add(kf)
                elif len(merge) < count:
                    merge.add(kf)
        return loop, merge


def flatten(bows, freqs=None):
    """keyframe-major arrays of the C ABI / the oracle, keyframes in ascending id order"""
    ids = sorted(bows)
    offs = np.zeros(len(ids) + 1, np.int64)
    words, vals, counts = [], [], []
    for n, k in enumerate(ids):
        for w in sorted(bows[k]):
            words.append(w)
            vals.append(bows[k][w])
            counts.append(1 if freqs is None else freqs[k][w])
        offs[n + 1] = len(words)
    return ids, offs, np.asarray(words, np.uint32), np.asarray(vals, np.float64), np.asarray(counts, np.uint32)
