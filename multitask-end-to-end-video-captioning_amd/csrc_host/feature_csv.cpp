// feature_csv.cpp -- one-pass reader of the reference's feature text files (include/s2vt_host.h).
// Format written by tf_feature_extract.py:153-154 and parsed, every run, into Python lists of STRINGS by
// get_video_feature_caption_pair (tf_s2vt.py:332-339): one line per frame, "vid<N>_frame_<k>,f0,f1,...,f<d-1>".
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/s2vt_host.h"

extern "C" {

int s2vt_feature_csv_scan(const char* path, int64_t* n_rows, int32_t* dim)
{
    if (!path || !n_rows || !dim) return -1;
    FILE* f = std::fopen(path, "rb");
    if (!f) return -2;
    std::vector<char> buf(1 << 20);
    int64_t rows = 0;
    int32_t d = -1, commas = 0;
    bool any = false;
    size_t got;
    while ((got = std::fread(buf.data(), 1, buf.size(), f)) > 0) {
        for (size_t i = 0; i < got; ++i) {
            const char c = buf[i];
            if (c == '\n') {
                if (any) {
                    if (d < 0) d = commas;
                    else if (d != commas) { std::fclose(f); return -3; }      // ragged rows
                    ++rows;
                }
                commas = 0; any = false;
            } else {
                if (c == ',') ++commas;
                if (c != '\r') any = true;
            }
        }
    }
    if (any) { if (d < 0) d = commas; else if (d != commas) { std::fclose(f); return -3; } ++rows; }
    std::fclose(f);
    *n_rows = rows;
    *dim = d < 0 ? 0 : d;
    return 0;
}

int s2vt_feature_csv_read(const char* path, int64_t n_rows, int32_t dim, float* out, char* ids, int32_t id_len)
{
    if (!path || !out || !ids || n_rows < 0 || dim <= 0 || id_len < 2) return -1;
    FILE* f = std::fopen(path, "rb");
    if (!f) return -2;
    std::string line;
    std::vector<char> buf(1 << 20);
    int64_t r = 0;
    auto flush = [&](std::string& s) -> int {
        while (!s.empty() && (s.back() == '\r' || s.back() == '\n')) s.pop_back();
        if (s.empty()) return 0;
        if (r >= n_rows) return -3;
        const char* p = s.c_str();
        const char* comma = std::strchr(p, ',');
        if (!comma) return -3;
        size_t n = (size_t)(comma - p);
        if (n >= (size_t)id_len) n = id_len - 1;
        std::memcpy(ids + (size_t)r * id_len, p, n);
        ids[(size_t)r * id_len + n] = 0;
        const char* q = comma + 1;
        float* o = out + (size_t)r * dim;
        for (int j = 0; j < dim; ++j) {
            char* e;
            o[j] = std::strtof(q, &e);
            if (e == q) return -3;
            q = (*e == ',') ? e + 1 : e;
        }
        ++r;
        s.clear();
        return 0;
    };
    size_t got;
    while ((got = std::fread(buf.data(), 1, buf.size(), f)) > 0) {
        size_t start = 0;
        for (size_t i = 0; i < got; ++i)
            if (buf[i] == '\n') {
                line.append(buf.data() + start, i - start);
                const int rc = flush(line);
                if (rc) { std::fclose(f); return rc; }
                start = i + 1;
            }
        line.append(buf.data() + start, got - start);
    }
    const int rc = flush(line);
    std::fclose(f);
    if (rc) return rc;
    return r == n_rows ? 0 : -3;
}

}  // extern "C"
