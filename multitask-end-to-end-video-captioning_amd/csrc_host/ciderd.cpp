// ciderd.cpp -- CIDEr-D reward on integer token ids (include/s2vt_host.h).  Host code, OpenMP over rows.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <new>
#include <unordered_map>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

#include "../../include/s2vt_host.h"

namespace {

constexpr int kN = 4;               // 1..4-grams
constexpr double kSigma = 6.0;

// an n-gram (n <= 4) of ids as a 128-bit key: 4 x (id + 1), 0 = absent slot
struct Key {
    uint64_t a, b;
    bool operator==(const Key& o) const { return a == o.a && b == o.b; }
};
struct KeyHash {
    size_t operator()(const Key& k) const
    {
        uint64_t x = k.a * 0x9E3779B97F4A7C15ull ^ (k.b + 0xC2B2AE3D27D4EB4Full + (k.a << 6) + (k.a >> 2));
        x ^= x >> 31; x *= 0xD6E8FEB86659FD93ull; x ^= x >> 32;
        return (size_t)x;
    }
};
inline Key make_key(const int32_t* w, int n)
{
    uint64_t v[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; ++i) v[i] = (uint64_t)(uint32_t)w[i] + 1;
    return Key{(v[0] << 32) | v[1], (v[2] << 32) | v[3]};
}
inline int key_len(const Key& k)
{
    return ((k.a >> 32) != 0) + ((k.a & 0xFFFFFFFFull) != 0) + ((k.b >> 32) != 0) + ((k.b & 0xFFFFFFFFull) != 0);
}

typedef std::unordered_map<Key, float, KeyHash> Counts;     // n-gram -> term frequency (all n in one map)

void count_ngrams(const int32_t* w, int len, Counts& c)
{
    for (int n = 1; n <= kN; ++n)
        for (int i = 0; i + n <= len; ++i) c[make_key(w + i, n)] += 1.0f;
}

// One video's references as an inverted index: the distinct n-grams of all its references, sorted, each with the
// list of (reference, tf-idf weight) pairs that contain it -- a candidate n-gram costs ONE binary search per video
// instead of one hash probe per reference, and scoring allocates nothing.
struct Posting { uint32_t ref; double w; };
struct VideoIndex {
    std::vector<Key> keys;                 // sorted
    std::vector<uint32_t> begin;           // keys.size() + 1 offsets into post
    std::vector<Posting> post;
    std::vector<double> norm;              // [n_refs][kN]
    std::vector<double> length;            // [n_refs]
};
inline bool key_less(const Key& x, const Key& y) { return x.a < y.a || (x.a == y.a && x.b < y.b); }

}  // namespace

struct s2vt_cider {
    int32_t n_videos;
    double ref_len;                                              // log(number of videos)
    std::unordered_map<Key, float, KeyHash> df;                  // document frequency (documents = videos)
    std::vector<VideoIndex> videos;

    double idf(const Key& k) const
    {
        auto it = df.find(k);
        return ref_len - std::log(std::max(1.0, it == df.end() ? 0.0 : (double)it->second));
    }
};

extern "C" {

s2vt_cider* s2vt_cider_create(const int32_t* tokens, const int64_t* offsets, const int32_t* video_of_ref, int32_t n_refs,
                              int32_t n_videos)
{
    if (!tokens || !offsets || !video_of_ref || n_refs <= 0 || n_videos <= 0) return nullptr;
    for (int r = 0; r < n_refs; ++r)
        if (video_of_ref[r] < 0 || video_of_ref[r] >= n_videos || offsets[r + 1] < offsets[r]) return nullptr;
    s2vt_cider* h = new (std::nothrow) s2vt_cider;
    if (!h) return nullptr;
    h->n_videos = n_videos;
    h->ref_len = std::log((double)n_videos);
    // per-reference counts, grouped by video
    std::vector<std::vector<Counts>> counts(n_videos);
    for (int r = 0; r < n_refs; ++r) {
        Counts c;
        count_ngrams(tokens + offsets[r], (int)(offsets[r + 1] - offsets[r]), c);
        counts[video_of_ref[r]].push_back(std::move(c));
    }
    // document frequency: an n-gram counts once per VIDEO whose references contain it
    for (int v = 0; v < n_videos; ++v) {
        std::unordered_map<Key, char, KeyHash> seen;
        for (const Counts& c : counts[v])
            for (const auto& kv : c) seen.emplace(kv.first, 1);
        for (const auto& kv : seen) h->df[kv.first] += 1.0f;
    }
    h->videos.resize(n_videos);
    for (int v = 0; v < n_videos; ++v) {
        VideoIndex& vi = h->videos[v];
        const size_t nr = counts[v].size();
        vi.norm.assign(nr * kN, 0.0);
        vi.length.assign(nr, 0.0);
        struct Item { Key k; uint32_t ref; double w; };
        std::vector<Item> items;
        for (size_t i = 0; i < nr; ++i)
            for (const auto& kv : counts[v][i]) {
                const int n = key_len(kv.first) - 1;
                const double g = (double)kv.second * h->idf(kv.first);
                items.push_back({kv.first, (uint32_t)i, g});
                vi.norm[i * kN + n] += g * g;
                if (n == 1) vi.length[i] += kv.second;
            }
        for (double& x : vi.norm) x = std::sqrt(x);
        std::sort(items.begin(), items.end(), [](const Item& x, const Item& y) {
            return key_less(x.k, y.k) || (x.k == y.k && x.ref < y.ref);
        });
        for (size_t i = 0; i < items.size(); ++i) {
            if (i == 0 || !(items[i].k == items[i - 1].k)) {
                vi.keys.push_back(items[i].k);
                vi.begin.push_back((uint32_t)vi.post.size());
            }
            vi.post.push_back({items[i].ref, items[i].w});
        }
        vi.begin.push_back((uint32_t)vi.post.size());
    }
    return h;
}

void s2vt_cider_destroy(s2vt_cider* h) { delete h; }

int32_t s2vt_cider_num_videos(const s2vt_cider* h) { return h ? h->n_videos : 0; }

int s2vt_cider_score(const s2vt_cider* h, const int32_t* ids, int32_t N, int32_t Tc, int32_t eos_id,
                     const int32_t* video_of_row, float* out, int32_t n_threads)
{
    if (!h || !ids || !video_of_row || !out || N < 0 || Tc <= 0) return -1;
    for (int n = 0; n < N; ++n)
        if (video_of_row[n] < 0 || video_of_row[n] >= h->n_videos) return -1;
    size_t max_refs = 0;
    for (const VideoIndex& vi : h->videos) max_refs = std::max(max_refs, vi.length.size());
    // ~30 index probes per caption (a 384-caption batch scores in ~3 ms on one core): threads only on request --
    // inside a process that also hosts torch's OpenMP pool a parallel region costs more than it saves
#ifdef _OPENMP
    const int nt = n_threads > 1 ? n_threads : 1;
#pragma omp parallel num_threads(nt) if (nt > 1)
#endif
    {
        std::vector<Key> keys((size_t)kN * Tc);             // per-thread scratch, allocated once per call
        std::vector<double> val(max_refs * kN);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (int r = 0; r < N; ++r) {
            const int32_t* w = ids + (size_t)r * Tc;
            int len = 0;
            while (len < Tc && w[len] != eos_id) ++len;             // words before the first <eos>
            // the candidate's n-grams, sorted (all n together; shorter n-grams have zero low slots and sort first)
            int nk = 0;
            for (int n = 1; n <= kN; ++n)
                for (int i = 0; i + n <= len; ++i) keys[nk++] = make_key(w + i, n);
            std::sort(keys.begin(), keys.begin() + nk, key_less);
            const VideoIndex& vi = h->videos[video_of_row[r]];
            const size_t nr = vi.length.size();
            std::fill(val.begin(), val.begin() + nr * kN, 0.0);
            double hnorm2[kN] = {0, 0, 0, 0}, hlen = 0.0;
            for (int i = 0; i < nk;) {
                int j = i + 1;
                while (j < nk && keys[j] == keys[i]) ++j;
                const Key k = keys[i];
                const int n = key_len(k) - 1;
                const double g = (double)(j - i) * h->idf(k);
                hnorm2[n] += g * g;
                if (n == 1) hlen += (double)(j - i);
                auto it = std::lower_bound(vi.keys.begin(), vi.keys.end(), k, key_less);
                if (it != vi.keys.end() && *it == k) {
                    const size_t ki = (size_t)(it - vi.keys.begin());
                    for (uint32_t q = vi.begin[ki]; q < vi.begin[ki + 1]; ++q)
                        val[(size_t)vi.post[q].ref * kN + n] += std::min(g, vi.post[q].w) * vi.post[q].w;
                }
                i = j;
            }
            double hnorm[kN];
            for (int n = 0; n < kN; ++n) hnorm[n] = std::sqrt(hnorm2[n]);
            double score[kN] = {0, 0, 0, 0};
            for (size_t i = 0; i < nr; ++i) {
                const double delta = hlen - vi.length[i];
                const double pen = std::exp(-(delta * delta) / (2.0 * kSigma * kSigma));
                for (int n = 0; n < kN; ++n) {
                    double v = val[i * kN + n];
                    if (hnorm[n] != 0.0 && vi.norm[i * kN + n] != 0.0) v /= hnorm[n] * vi.norm[i * kN + n];
                    score[n] += v * pen;
                }
            }
            double avg = (score[0] + score[1] + score[2] + score[3]) / kN;
            if (nr) avg /= (double)nr;
            out[r] = (float)(avg * 10.0);
        }
    }
    return 0;
}

}  // extern "C"
