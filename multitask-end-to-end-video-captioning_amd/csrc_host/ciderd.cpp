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

struct Vec {
    std::vector<std::pair<Key, double>> g[kN];   // tf-idf weights per n
    double norm[kN];
    double length;                               // number of bigrams (the published scorer's "length")
};

}  // namespace

struct s2vt_cider {
    int32_t n_videos;
    double ref_len;                                              // log(number of videos)
    std::unordered_map<Key, float, KeyHash> df;                  // document frequency (documents = videos)
    std::vector<std::vector<Vec>> refs;                          // per video: one Vec per reference
    std::vector<std::vector<std::unordered_map<Key, double, KeyHash>>> ref_maps;   // same weights, hashed, per n merged

    void to_vec(const Counts& c, Vec& v) const
    {
        for (int n = 0; n < kN; ++n) { v.g[n].clear(); v.norm[n] = 0.0; }
        v.length = 0.0;
        for (const auto& kv : c) {
            const int n = key_len(kv.first) - 1;
            auto it = df.find(kv.first);
            const double d = std::log(std::max(1.0, it == df.end() ? 0.0 : (double)it->second));
            const double g = (double)kv.second * (ref_len - d);
            v.g[n].push_back({kv.first, g});
            v.norm[n] += g * g;
            if (n == 1) v.length += kv.second;
        }
        for (int n = 0; n < kN; ++n) v.norm[n] = std::sqrt(v.norm[n]);
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
    h->refs.resize(n_videos);
    h->ref_maps.resize(n_videos);
    for (int v = 0; v < n_videos; ++v) {
        h->refs[v].resize(counts[v].size());
        h->ref_maps[v].resize(counts[v].size());
        for (size_t i = 0; i < counts[v].size(); ++i) {
            h->to_vec(counts[v][i], h->refs[v][i]);
            auto& m = h->ref_maps[v][i];
            for (int n = 0; n < kN; ++n)
                for (const auto& kg : h->refs[v][i].g[n]) m.emplace(kg.first, kg.second);
        }
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
#ifdef _OPENMP
    const int nt = n_threads > 0 ? n_threads : omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 8) num_threads(nt)
#endif
    for (int r = 0; r < N; ++r) {
        const int32_t* w = ids + (size_t)r * Tc;
        int len = 0;
        while (len < Tc && w[len] != eos_id) ++len;                 // words before the first <eos>
        Counts c;
        count_ngrams(w, len, c);
        Vec hv;
        h->to_vec(c, hv);
        const auto& refs = h->refs[video_of_row[r]];
        const auto& maps = h->ref_maps[video_of_row[r]];
        double score[kN] = {0, 0, 0, 0};
        for (size_t i = 0; i < refs.size(); ++i) {
            const Vec& rv = refs[i];
            const double delta = hv.length - rv.length;
            const double pen = std::exp(-(delta * delta) / (2.0 * kSigma * kSigma));
            for (int n = 0; n < kN; ++n) {
                double val = 0.0;
                for (const auto& kg : hv.g[n]) {
                    auto it = maps[i].find(kg.first);
                    if (it != maps[i].end()) val += std::min(kg.second, it->second) * it->second;
                }
                if (hv.norm[n] != 0.0 && rv.norm[n] != 0.0) val /= hv.norm[n] * rv.norm[n];
                score[n] += val * pen;
            }
        }
        double avg = (score[0] + score[1] + score[2] + score[3]) / kN;
        if (!refs.empty()) avg /= (double)refs.size();
        out[r] = (float)(avg * 10.0);
    }
    return 0;
}

}  // extern "C"
