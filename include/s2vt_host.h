/*
 * s2vt_host.h -- C ABI of libs2vt_host.so: the HOST-side caller work around the device hot path
 * (SURVEY.md section 8(f) rank 1): the self-critical reward.  CPU only (g++ / OpenMP), no HIP.
 *
 * Replaces, for the REINFORCE step, evaluate_captions_cider (cider_evaluation.py:60-87), which calls
 * pyciderevalcap.ciderD.CiderD(df='msvd').compute_score on Python strings, once for the K*B sampled
 * captions and once for the B greedy captions of every step, after an O(#captions) scan per video for
 * its references (get_captions, reinforcement_multisampling_tf_s2vt.py:600-601).  Here the references
 * are tokenised ONCE into integer ids, their TF-IDF vectors and norms are precomputed per video, and a
 * step scores [N, Tc] int32 token-id rows straight from the sampler (no id -> string -> n-gram round
 * trip), multi-threaded over rows.
 *
 * Algorithm: CIDEr-D of Vedantam et al. as published in pyciderevalcap/ciderD/ciderD_scorer.py
 * (the third-party dependency is NOT vendored by the reference and is absent here; its DF pickle
 * 'msvd' is absent too -- the DF table is built from the reference corpus given to s2vt_cider_create,
 * one document per video, as the scorer's own "corpus" mode does.  Parity with the pickle: unpinned.)
 *   n = 1..4 grams, tf-idf g = tf * (log(n_videos) - log(max(1, df))), clipped-count cosine with the
 *   Gaussian length penalty exp(-(len_c - len_r)^2 / (2 * 6^2)) (len = number of BIGRAMS, as the
 *   scorer counts it), mean over n and over the video's references, x 10.
 */
#ifndef S2VT_HOST_H
#define S2VT_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct s2vt_cider s2vt_cider;

/* References: n_refs sentences as one int32 token stream; sentence r = tokens[offsets[r] .. offsets[r+1]) and
 * belongs to video video_of_ref[r] in [0, n_videos).  Token ids are arbitrary non-negative int32 (words outside
 * the model vocabulary get ids >= |V| from the caller, so that they count in n-grams and DF but never match).
 * Returns NULL on bad arguments. */
s2vt_cider* s2vt_cider_create(const int32_t* tokens, const int64_t* offsets, const int32_t* video_of_ref, int32_t n_refs,
                              int32_t n_videos);
void s2vt_cider_destroy(s2vt_cider* h);

/* Score N candidate rows of Tc token ids (decode_captions semantics, cider_evaluation.py:122-143: the caption is
 * the ids BEFORE the first eos_id; an empty caption scores against the references like any other) against the
 * references of video_of_row[n].  out[n] = CIDEr-D * 10 as the reference's reward.  n_threads <= 1: the calling thread (a 384-caption batch takes ~3 ms); > 1: OpenMP threads.
 * Returns 0, or -1 on bad arguments (a video index out of range included). */
int s2vt_cider_score(const s2vt_cider* h, const int32_t* ids, int32_t N, int32_t Tc, int32_t eos_id,
                     const int32_t* video_of_row, float* out, int32_t n_threads);

int32_t s2vt_cider_num_videos(const s2vt_cider* h);

/* ---- feature files ------------------------------------------------------
 * The reference stores Inception-ResNet-v2 pool features as TEXT, one line per frame,
 * "vid<N>_frame_<k>,f0,...,f<d-1>" (writer tf_feature_extract.py:153-154), and re-parses the whole file into
 * Python lists of strings on every run (tf_s2vt.py:332-339).  These two calls read such a file once into a
 * contiguous float32 [n_rows, dim] block (+ fixed-width, NUL-terminated row ids) so that the caller can cache
 * it as .npy and hand the device coalesced [B, Tv, d] batches.
 * scan: rows and floats per row (-2 cannot open, -3 ragged rows).  read: fills out / ids (id_len bytes each). */
int s2vt_feature_csv_scan(const char* path, int64_t* n_rows, int32_t* dim);
int s2vt_feature_csv_read(const char* path, int64_t n_rows, int32_t dim, float* out, char* ids, int32_t id_len);

#ifdef __cplusplus
}
#endif
#endif
