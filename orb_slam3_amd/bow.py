"""Host-side mirror of the vocabulary part of the path: DBoW2Handler::Precompute
(src/features/handlers/DBoW2/dbo_w2_handler.cpp:74-93) = TemplatedVocabulary<FORB>::transform(features, BowVector,
FeatureVector, levelsup) (DBoW2/DBoW2/TemplatedVocabulary.h:1125-1189).  The per-feature tree descent (:1218-1259)
runs on the GPU (orbgpu_bow_transform); the BowVector / FeatureVector bookkeeping below is the reference's std::map
arithmetic in double, kept on the host.  There is no CPU descent here: without the HIP library this raises."""
import ctypes as C
import math

import numpy as np

from . import _lib

TF_IDF, TF, IDF, BINARY = 0, 1, 2, 3          # DBoW2::WeightingType, BowVector.h
L1_NORM, L2_NORM, CHI_SQUARE, KL, BHATTACHARYYA, DOT_PRODUCT = 0, 1, 2, 3, 4, 5   # DBoW2::ScoringType


class Vocabulary:
    """Flat vocabulary tree: node 0 is the root; children[child_off[v]:child_off[v+1]] are node v's children in
    Node::children order; node_desc[v] its 32-byte descriptor; word_id[v] / weight[v] are Node::word_id / Node::weight
    (meaningful for leaves); levels = m_L."""

    def __init__(self, extractor, child_off, children, node_desc, word_id, weight, levels, weighting=TF_IDF,
                 scoring=L1_NORM):
        self._fe = extractor
        self._lib = extractor._lib
        self.child_off = np.ascontiguousarray(child_off, np.int32)
        self.children = np.ascontiguousarray(children, np.uint32)
        self.node_desc = np.ascontiguousarray(node_desc, np.uint8).reshape(-1, 32)
        self.word_id = np.ascontiguousarray(word_id, np.uint32)
        self.weight = np.ascontiguousarray(weight, np.float64)
        self.levels = int(levels)
        self.weighting, self.scoring = weighting, scoring
        h = C.c_void_p()
        _lib.check(self._lib.orbgpu_vocab_create(extractor.handle, len(self.node_desc), _lib.ptr(self.child_off),
                                                 _lib.ptr(self.children), _lib.ptr(self.node_desc), _lib.ptr(self.word_id),
                                                 self.levels, C.byref(h)), extractor.handle)
        self._h = h
        leaves = np.nonzero(np.diff(self.child_off) == 0)[0]
        self._leaf_of_word = {int(self.word_id[v]): int(v) for v in leaves}

    def close(self):
        if self._h:
            self._lib.orbgpu_vocab_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def transform_ids(self, descriptors, levelsup=4):
        """(word id, node id at level L - levelsup) per descriptor: the GPU tree descent."""
        d = np.ascontiguousarray(descriptors, np.uint8).reshape(-1, 32)
        w = np.zeros(max(len(d), 1), np.uint32)
        nd = np.zeros(max(len(d), 1), np.uint32)
        _lib.check(self._lib.orbgpu_bow_transform(self._fe.handle, self._h, _lib.ptr(d), len(d), int(levelsup), _lib.ptr(w),
                                                  _lib.ptr(nd)), self._fe.handle)
        return w[:len(d)], nd[:len(d)]

    def _must_normalize(self):
        """GeneralScoring::mustNormalize, ScoringObject.h: L1/L2/CHI/KL/BHATTACHARYYA normalise, DOT_PRODUCT does not."""
        if self.scoring == DOT_PRODUCT:
            return False, None
        return True, ("L2" if self.scoring == L2_NORM else "L1")

    def transform(self, descriptors, levelsup=4):
        """-> (BowVector {word id: value}, FeatureVector {node id: [feature ids]}), TemplatedVocabulary.h:1125-1189."""
        words, nodes = self.transform_ids(descriptors, levelsup)
        v, fv = {}, {}
        must, norm = self._must_normalize()
        tf_like = self.weighting in (TF_IDF, TF)
        for i in range(len(words)):
            wid = int(words[i])
            w = float(self.weight[self._leaf_of_word[wid]])
            if w > 0:                                   # not stopped
                if tf_like:
                    v[wid] = v.get(wid, 0.0) + w        # BowVector::addWeight
                else:
                    v.setdefault(wid, w)                # BowVector::addIfNotExist
                fv.setdefault(int(nodes[i]), []).append(i)   # FeatureVector::addFeature
        v = dict(sorted(v.items()))
        if tf_like and v and not must:
            nd = float(len(v))
            for k in v:
                v[k] /= nd
        if must:                                        # BowVector::normalize, BowVector.cpp:61-84
            if norm == "L1":
                s = 0.0
                for k in v:
                    s += math.fabs(v[k])
            else:
                s = 0.0
                for k in v:
                    s += v[k] * v[k]
                s = math.sqrt(s)
            if s > 0.0:
                for k in v:
                    v[k] /= s
        return v, dict(sorted(fv.items()))
