"""Host-side mirror of the reference's keyframe database, frame::DBoW2Database
(src/frame/database/DBoW2/dbo_w2_database.{h,cpp}) behind IKeyFrameDatabase (src/frame/database/ikey_frame_database.h).

The inverted file (word -> {keyframe: count}) and its binary stream (Serialize :39-51, stream constructor :16-32) are
kept on the host as the reference keeps them.  The query -- SearchWordSharingKeyFrames (:83-103) plus the
vocabulary->score calls of FilterCandidatesByWordSharingAcceptableScore (:114-132) -- is ONE GPU call over all
keyframes (orbgpu_bow_score); the 0.8 * max cut, and the covisibility accumulation of FilterCandidatesByCovisibility
(:134-179), which walks the caller's covisibility graph, stay on the host.  There is no CPU scoring path here:
without the HIP library this raises."""
import ctypes as C
import struct

import numpy as np

from . import _lib
from .bow import L1_NORM

_SIZE_MAX = (1 << 64) - 1


class DBoW2Database:
    """Keyframes are named by their id (KeyFrame::Id(), the value the stream holds); where the reference iterates an
    unordered container of KeyFrame* this class iterates in ascending id order."""

    def __init__(self, extractor, vocabulary_size, scoring=L1_NORM, with_counts=False):
        self._fe = extractor
        self._lib = extractor._lib
        self.scoring = int(scoring)
        self.with_counts = bool(with_counts)
        self.inverted_file = [dict() for _ in range(int(vocabulary_size))]   # DBoW2Database(vocabulary), :12-14
        self._bow = {}            # keyframe id -> BowVector {word: value} (the reference reads it from the handler)
        self._h = None            # device-resident copy (orbgpu_bowdb), created with the first keyframe

    def _handle(self):
        if self._h is None:
            h = C.c_void_p()
            _lib.check(self._lib.orbgpu_bowdb_create(self._fe.handle, self.scoring, int(self.with_counts), C.byref(h)),
                       self._fe.handle)
            self._h = h
        return self._h

    def close(self):
        if self._h:
            self._lib.orbgpu_bowdb_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _upload(self, kf_id):
        bow = self._bow[kf_id]
        words = np.fromiter(bow.keys(), np.uint32, len(bow))
        values = np.fromiter(bow.values(), np.float64, len(bow))
        counts = None
        if self.with_counts:      # a BowVector word the inverted file does not hold for this keyframe counts 0
            counts = np.asarray([self.inverted_file[w].get(kf_id, 0) for w in bow], np.uint32)
        _lib.check(self._lib.orbgpu_bowdb_append(self._handle(), kf_id, _lib.ptr(words), _lib.ptr(values), _lib.ptr(counts),
                                                 len(words)), self._fe.handle)

    # ---- IKeyFrameDatabase ------------------------------------------------------------------------
    def Append(self, kf_id, bow_vector, word_frequencies=None):
        """:53-61.  word_frequencies = DBoW2Handler::GetWordFrequencies (one count per word of the BowVector,
        dbo_w2_handler.cpp:90-92); None = 1 per word, which is what Precompute produces."""
        kf_id = int(kf_id)
        wf = {int(w): 1 for w in bow_vector} if word_frequencies is None else {int(w): int(c) for w, c in word_frequencies.items()}
        if not self.with_counts and (set(wf) != {int(w) for w in bow_vector} or any(c != 1 for c in wf.values())):
            raise ValueError("word frequencies other than 1 per BowVector word need with_counts=True")
        for w in wf:
            if not 0 <= w < len(self.inverted_file):
                raise ValueError(f"word {w} outside the vocabulary")     # the reference asserts, :58
        for w, c in wf.items():
            self.inverted_file[w][kf_id] = c
        self._bow[kf_id] = {int(w): float(v) for w, v in sorted(bow_vector.items())}
        if self._lib is not None:
            self._upload(kf_id)

    def Erase(self, kf_id):
        """:187-195: the keyframe leaves the maps of the words of its BowVector."""
        kf_id = int(kf_id)
        for w in self._bow.get(kf_id, {}):
            self.inverted_file[w].pop(kf_id, None)
        self._bow.pop(kf_id, None)
        if self._h:
            _lib.check(self._lib.orbgpu_bowdb_erase(self._h, kf_id), self._fe.handle)

    def Serialize(self, stream):
        """:39-51: u64 word count; per word u64 entry count, then (u64 keyframe id, u64 count) per entry.  The
        reference writes a word's entries in KeyFrame* order (allocator dependent); here ascending id."""
        stream.write(struct.pack("<Q", len(self.inverted_file)))
        for inv_map in self.inverted_file:
            stream.write(struct.pack("<Q", len(inv_map)))
            for kf_id in sorted(inv_map):
                stream.write(struct.pack("<QQ", kf_id, inv_map[kf_id]))

    @classmethod
    def Deserialize(cls, stream, extractor, bow_vectors, scoring=L1_NORM):
        """Stream constructor, :16-32.  The stream holds only the inverted file; the BowVectors come from the
        keyframes' handlers (bow_vectors: {keyframe id: {word: value}}), as in the reference."""
        def rd():
            b = stream.read(8)
            if len(b) != 8:
                raise ValueError("truncated keyframe database stream")
            return struct.unpack("<Q", b)[0]
        n_words = rd()
        maps = []
        for _ in range(n_words):
            m = {}
            for _ in range(rd()):
                kf_id = rd()
                m[kf_id] = rd()
            maps.append(m)
        seen = {kf for m in maps for kf in m}
        missing = seen - set(bow_vectors)
        if missing:
            raise ValueError(f"no BowVector for keyframes {sorted(missing)[:5]}")
        db = cls(extractor, n_words, scoring, with_counts=any(c != 1 for m in maps for c in m.values()))
        db.inverted_file = maps
        db._bow = {int(k): {int(w): float(v) for w, v in sorted(bow_vectors[k].items())} for k in sorted(seen)}
        if db._lib is not None:
            for k in db._bow:
                db._upload(k)
        return db

    # ---- the query ----------------------------------------------------------------------------------
    def ScoreAll(self, bow_vector, word_frequencies=None):
        """-> (ids, shared word counts u64, scores f64) over every keyframe: one orbgpu_bowdb_query."""
        n = len(self._bow)
        q = sorted((int(w), float(v)) for w, v in bow_vector.items())
        qw = np.asarray([w for w, _ in q], np.uint32)
        qv = np.asarray([v for _, v in q], np.float64)
        qc = None if word_frequencies is None else np.asarray([int(word_frequencies.get(w, 0)) for w, _ in q], np.uint32)
        ids = np.zeros(max(n, 1), np.uint64)
        shared = np.zeros(max(n, 1), np.uint64)
        score = np.zeros(max(n, 1), np.float64)
        n_out = C.c_int32(0)
        _lib.check(self._lib.orbgpu_bowdb_query(self._handle(), _lib.ptr(qw), _lib.ptr(qv), _lib.ptr(qc), len(qw), n,
                                                _lib.ptr(ids), _lib.ptr(shared), _lib.ptr(score), C.byref(n_out)),
                   self._fe.handle)
        assert n_out.value == n
        order = np.argsort(ids[:n], kind="stable")        # ascending id: the iteration order of this mirror
        return [int(k) for k in ids[:n][order]], shared[:n][order], score[:n][order]

    def SearchWordSharingKeyFrames(self, bow_vector, word_frequencies=None):
        """:83-103 -> {keyframe id: shared word count} (keyframes sharing no word are absent, as in the reference)."""
        ids, shared, _ = self.ScoreAll(bow_vector, word_frequencies)
        return {k: int(s) for k, s in zip(ids, shared) if s}

    def DetectNBestCandidates(self, bow_vector, covisible=None, map_of=None, current_map=None, count=_SIZE_MAX,
                              word_frequencies=None):
        """:63-81 -> (loop candidates, merge candidates) as sets of keyframe ids.
        covisible(kf_id) -> ids of its (at most 10) best covisible keyframes (CovisibilityGraph::GetCovisibleKeyFrames(10),
        :146); map_of(kf_id) -> the map the keyframe belongs to (:173).  Both default to "none" / "current map"."""
        ids, shared, score = self.ScoreAll(bow_vector, word_frequencies)
        word_sharing = {k: int(s) for k, s in zip(ids, shared) if s}
        if not word_sharing:                                             # :72
            return set(), set()
        # FilterCandidatesByWordSharingAcceptableScore, :114-132
        max_sharing = max(word_sharing.values())
        min_common = int(0.8 * max_sharing)                              # size_t min_common_words = 0.8 * max
        score_of = dict(zip(ids, (float(s) for s in score)))
        reloc = {k: score_of[k] for k, s in word_sharing.items() if s > min_common}
        # FilterCandidatesByCovisibility, :134-179
        acc, best_acc = {}, 0.0
        for kf, sc in reloc.items():
            best_score, acc_score, best_kf = sc, sc, kf
            for nb in (covisible(kf) if covisible else ()):
                if nb not in word_sharing:
                    continue
                if nb in reloc:
                    acc_score += reloc[nb]
                    if reloc[nb] > best_score:
                        best_kf, best_score = nb, reloc[nb]
            acc[best_kf] = acc_score
            if acc_score > best_acc:
                best_acc = acc_score
        min_acceptable = 0.75 * best_acc
        loop, merge = set(), set()
        for kf, sc in acc.items():
            if sc > min_acceptable:
                same_map = True if map_of is None else map_of(kf) == current_map
                if same_map and len(loop) < count:
                    loop.add(kf)
                elif len(merge) < count:
                    merge.add(kf)
        return loop, merge

    def DetectRelocCandidates(self, bow_vector, covisible=None, map_of=None, current_map=None, word_frequencies=None):
        """:181-185"""
        return self.DetectNBestCandidates(bow_vector, covisible, map_of, current_map, _SIZE_MAX, word_frequencies)[0]
