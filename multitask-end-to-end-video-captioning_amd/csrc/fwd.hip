// fwd.hip -- instantiations + shape dispatch of the forward contraction kernel (gemm_mfma.h).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "internal.h"

namespace s2vt {

namespace {

typedef void (*KernelFn)(const GemmArgs);

struct CfgEntry {
    const char* name;
    KernelFn vec, scalar;   // 16-byte vector loads / scalar loads (unaligned or odd shapes)
    int BM, CG, NT, lds_bytes;
    KernelFn vec_om, scalar_om;   // the live-row forms (GemmArgs::omap / m_dev): cell-step and pick tiles only
    int fallback;                 // LDS-DMA entries (vector path only): the table index of the register-staged tile that takes the launch
                                  // when the operands are not 16-byte aligned (-1: this entry has a scalar form of its own)
};

template <int WM, int WN, int TM, int TN, int NG, int EPI, int BKT = 32, int PW = 0, bool BT = false>
constexpr CfgEntry make_entry(const char* name)
{
    using C = GemmCfg<WM, WN, TM, TN, NG, EPI, true, BKT, PW, BT>;
    KernelFn om = nullptr, om_s = nullptr;
    if constexpr (EPI == EPI_LSTM || EPI == EPI_LSTM_GW || EPI == EPI_PICK) {
        om = gemm_kernel<WM, WN, TM, TN, NG, EPI, true, BKT, PW, BT, true>;
        om_s = gemm_kernel<WM, WN, TM, TN, NG, EPI, false, BKT, PW, BT, true>;
    }
    return CfgEntry{name, gemm_kernel<WM, WN, TM, TN, NG, EPI, true, BKT, PW, BT>,
                    gemm_kernel<WM, WN, TM, TN, NG, EPI, false, BKT, PW, BT>, C::BM, C::CG, C::NT, C::LDS_FLOATS * 4, om, om_s, -1};
}
// LDS-DMA ring tiles (gemm_mfma.h, DM > 0): PW loader waves issue buffer_load ... lds pieces DM stages deep, the MFMA waves only read
// fragments and multiply.  Vector path only; `fallback` names the register-staged tile for unaligned operands.
template <int WM, int WN, int TM, int TN, int NG, int EPI, int BKT, int PW, int DM, bool BT = false>
constexpr CfgEntry make_dma_entry(const char* name, int fallback)
{
    using C = GemmCfg<WM, WN, TM, TN, NG, EPI, true, BKT, PW, BT, DM>;
    KernelFn om = nullptr;
    if constexpr (EPI == EPI_LSTM || EPI == EPI_LSTM_GW || EPI == EPI_PICK) om = gemm_kernel<WM, WN, TM, TN, NG, EPI, true, BKT, PW, BT, true, DM>;
    return CfgEntry{name, gemm_kernel<WM, WN, TM, TN, NG, EPI, true, BKT, PW, BT, false, DM>, nullptr, C::BM, C::CG, C::NT, C::LDS_FLOATS * 4, om,
                    nullptr, fallback};
}

// name = BMxBN(wavesMxwavesN)
const CfgEntry kStore[] = {
    make_entry<4, 1, 1, 4, 1, EPI_STORE>("64x64(4x1)"),
    make_entry<2, 2, 2, 4, 1, EPI_STORE>("64x128(2x2)"),
    make_entry<2, 2, 4, 4, 1, EPI_STORE>("128x128(2x2)"),
    make_entry<4, 1, 2, 2, 1, EPI_STORE>("128x32(4x1)"),
    make_entry<4, 1, 1, 2, 1, EPI_STORE>("64x32(4x1)"),
    make_entry<2, 2, 2, 3, 1, EPI_STORE>("64x96(2x2)"),
    make_entry<2, 2, 3, 3, 1, EPI_STORE>("96x96(2x2)"),
    make_entry<2, 2, 3, 4, 1, EPI_STORE>("96x128(2x2)"),
    // LDS-DMA ring (round 6): 4 loader waves, 2 stages
    make_dma_entry<2, 2, 4, 4, 1, EPI_STORE, 32, 4, 2>("128x128(2x2)+4dma2", 2),
    make_dma_entry<2, 2, 3, 3, 1, EPI_STORE, 32, 4, 2>("96x96(2x2)+4dma2", 6),
    make_dma_entry<2, 2, 3, 3, 1, EPI_STORE, 32, 4, 3>("96x96(2x2)+4dma3", 6),
    make_dma_entry<2, 2, 4, 4, 1, EPI_STORE, 32, 2, 2>("128x128(2x2)+2dma2", 2),
};
// the same tiles for C = A . W^T with W given as [N][K] (backward data-gradient products): index-compatible with kStore
const CfgEntry kStoreNT[] = {
    make_entry<4, 1, 1, 4, 1, EPI_STORE, 32, 0, true>("nt64x64(4x1)"),
    make_entry<2, 2, 2, 4, 1, EPI_STORE, 32, 0, true>("nt64x128(2x2)"),
    make_entry<2, 2, 4, 4, 1, EPI_STORE, 32, 0, true>("nt128x128(2x2)"),
    make_entry<4, 1, 2, 2, 1, EPI_STORE, 32, 0, true>("nt128x32(4x1)"),
    make_entry<4, 1, 1, 2, 1, EPI_STORE, 32, 0, true>("nt64x32(4x1)"),
    make_entry<2, 2, 2, 3, 1, EPI_STORE, 32, 0, true>("nt64x96(2x2)"),
    make_entry<2, 2, 3, 3, 1, EPI_STORE, 32, 0, true>("nt96x96(2x2)"),
    make_entry<2, 2, 3, 4, 1, EPI_STORE, 32, 0, true>("nt96x128(2x2)"),
    make_dma_entry<2, 2, 4, 4, 1, EPI_STORE, 32, 4, 2, true>("nt128x128(2x2)+4dma2", 2),
    make_dma_entry<2, 2, 3, 3, 1, EPI_STORE, 32, 4, 2, true>("nt96x96(2x2)+4dma2", 6),
    make_dma_entry<2, 2, 3, 3, 1, EPI_STORE, 32, 4, 3, true>("nt96x96(2x2)+4dma3", 6),
    make_dma_entry<2, 2, 4, 4, 1, EPI_STORE, 32, 2, 2, true>("nt128x128(2x2)+2dma2", 2),
};
const CfgEntry kLstm[] = {
    // all four gates of 16 (32) units in one wave
    make_entry<4, 1, 1, 4, 4, EPI_LSTM>("64x16u(4x1)"),
    make_entry<2, 2, 2, 4, 4, EPI_LSTM>("64x32u(2x2)"),
    make_entry<2, 1, 1, 4, 4, EPI_LSTM>("32x16u(2x1)"),
    // gate-per-wave: rows x 16 units per workgroup, (row groups x 4 gate waves)
    make_entry<1, 4, 1, 1, 4, EPI_LSTM_GW>("gw16x16u(1x4)"),
    make_entry<1, 4, 2, 1, 4, EPI_LSTM_GW>("gw32x16u(1x4)"),
    make_entry<1, 4, 3, 1, 4, EPI_LSTM_GW>("gw48x16u(1x4)"),
    make_entry<1, 4, 4, 1, 4, EPI_LSTM_GW>("gw64x16u(1x4)"),
    make_entry<1, 4, 5, 1, 4, EPI_LSTM_GW>("gw80x16u(1x4)"),
    make_entry<2, 4, 3, 1, 4, EPI_LSTM_GW>("gw96x16u(2x4)"),
    make_entry<1, 4, 1, 1, 4, EPI_LSTM_GW, 64>("gw16x16u(1x4)k64"),
    // + 4 loader waves (the MFMA waves issue no loads / LDS stores): 3-5 % over the plain tiles at M = 320 / 384
    make_entry<1, 4, 5, 1, 4, EPI_LSTM_GW, 32, 4>("gw80x16u(1x4)+4"),
    make_entry<1, 4, 6, 1, 4, EPI_LSTM_GW, 32, 4>("gw96x16u(1x4)+4"),
    // LDS-DMA ring (round 6): 4 loader waves x DM stages; M <= 64: 16-row tiles, 2 / 4 loader waves, 8 / 5 stages deep
    make_dma_entry<1, 4, 6, 1, 4, EPI_LSTM_GW, 32, 4, 4>("gw96x16u(1x4)+4dma4", 11),
    make_dma_entry<1, 4, 1, 1, 4, EPI_LSTM_GW, 32, 2, 8>("gw16x16u(1x4)+2dma8", 9),
    make_dma_entry<1, 4, 1, 1, 4, EPI_LSTM_GW, 64, 4, 5>("gw16x16u(1x4)k64+4dma5", 9),
    make_dma_entry<1, 4, 2, 1, 4, EPI_LSTM_GW, 32, 4, 6>("gw32x16u(1x4)+4dma6", 4),
    make_dma_entry<1, 4, 4, 1, 4, EPI_LSTM_GW, 32, 4, 4>("gw64x16u(1x4)+4dma4", 6),
    make_dma_entry<1, 4, 6, 1, 4, EPI_LSTM_GW, 64, 4, 3>("gw96x16u(1x4)k64+4dma3", 11),
    make_dma_entry<1, 4, 6, 1, 4, EPI_LSTM_GW, 32, 4, 6>("gw96x16u(1x4)+4dma6", 11),
};
constexpr int kLstmGw80L = 10, kLstmGw96L = 11;
constexpr int kLstmGw96D = 12, kLstmGw16D = 13, kLstmGw16k64D = 14, kLstmGw32D = 15, kLstmGw64D = 16, kLstmGw96D6 = 18;
constexpr int kLstmGw16k64 = 9;     // 64-deep chunks for the M <= 64 step: 8 MFMAs per wave per 32-deep chunk leave the barrier dominant
constexpr int kLstmGwFirst = 3;      // index of gw16x16u; the gw entries follow in order of rows
const CfgEntry kPick[] = {
    make_entry<4, 1, 1, 4, 1, EPI_PICK>("64x64(4x1)"),
    make_entry<2, 2, 4, 4, 1, EPI_PICK>("128x128(2x2)"),
    make_entry<2, 2, 2, 4, 1, EPI_PICK>("64x128(2x2)"),
    make_entry<2, 4, 3, 3, 1, EPI_PICK>("96x192(2x4)"),
    make_entry<2, 2, 2, 3, 1, EPI_PICK>("64x96(2x2)"),
    make_entry<2, 2, 3, 2, 1, EPI_PICK>("96x64(2x2)"),
    make_entry<2, 2, 1, 3, 1, EPI_PICK>("32x96(2x2)"),
    // LDS-DMA ring (round 6)
    make_dma_entry<2, 2, 2, 3, 1, EPI_PICK, 32, 2, 2>("64x96(2x2)+2dma2", 4),
    make_dma_entry<2, 2, 2, 3, 1, EPI_PICK, 32, 4, 2>("64x96(2x2)+4dma2", 4),
    make_dma_entry<2, 2, 1, 3, 1, EPI_PICK, 32, 4, 6>("32x96(2x2)+4dma6", 6),
    make_dma_entry<2, 2, 1, 3, 1, EPI_PICK, 32, 2, 6>("32x96(2x2)+2dma6", 6),
    make_dma_entry<2, 2, 3, 3, 1, EPI_PICK, 32, 2, 3>("96x96(2x2)+2dma3", 4),
    make_dma_entry<2, 2, 3, 3, 1, EPI_PICK, 32, 4, 3>("96x96(2x2)+4dma3", 4),
    make_dma_entry<2, 2, 1, 3, 1, EPI_PICK, 64, 4, 4>("32x96(2x2)k64+4dma4", 6),
    make_dma_entry<2, 2, 3, 3, 1, EPI_PICK, 64, 4, 2>("96x96(2x2)k64+4dma2", 4),
};
constexpr int kPick64x96D2 = 7, kPick64x96D4 = 8, kPick32x96D4 = 9, kPick32x96D2 = 10;
// vocab pick, tile by M (tools/tune_pick_m.py on MI355X, H = 1000, |V| = 12000; us per launch 64x96 / 64x64 / 32x96):
// M=64: 62/51/49, 128: 63/57/58, 192: 76/75/67, 256: 78/76/76, 320: 102/90/92, 384: 101/110/103.  Below M ~ 128 every
// tile sits on a ~50 us floor (one workgroup per CU streaming its 192-256 KB slice of W and of the state, latency-bound).
constexpr int kPick64x96 = 4, kPick64x64 = 0, kPick32x96 = 6;
// S2VT_DMA (default 1): the LDS-DMA ring tiles where they exist for the shape; 0 = the register-staged tiles of round 5 (A/B switch)
bool dma_tiles()
{
    static const int on = [] { const char* e = getenv("S2VT_DMA"); return e ? atoi(e) : 1; }();
    return on != 0;
}
int choose_pick(int M)
{
    // (tools/tune_dma.py, round 6: no LDS-DMA pick tile beats its register-staged twin -- 64x96 needs three workgroups per CU, which leaves
    //  the ring two stages, one chunk of look-ahead; the 32x96 tile at <= 64 rows is bound by its 32 barriers, not by its loads)
    return M <= 256 ? kPick32x96 : (M <= 352 ? kPick64x64 : kPick64x96);
}

const CfgEntry* table(int epi, int* n)
{
    switch (epi) {
        case EPI_STORE: *n = sizeof(kStore) / sizeof(kStore[0]); return kStore;
        case EPI_STORE_NT: *n = sizeof(kStoreNT) / sizeof(kStoreNT[0]); return kStoreNT;
        case EPI_LSTM: *n = sizeof(kLstm) / sizeof(kLstm[0]); return kLstm;
        default: *n = sizeof(kPick) / sizeof(kPick[0]); return kPick;
    }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

bool can_vec(const GemmArgs& a, bool bt)
{
    if (!aligned16(a.W) || (a.ldw & 3) || (!bt && ((a.N & 3) || (a.gstride & 3)))) return false;
    if (bt)
        for (int s = 0; s < a.nseg; ++s)
            if (a.seg[s].kw & 3) return false;           // W^T form: kw is a column offset of 16-byte loads
    for (int s = 0; s < a.nseg; ++s) {
        const ASeg& sg = a.seg[s];
        if (!sg.ptr || sg.k <= 0) continue;
        if (!aligned16(sg.ptr) || (sg.ld & 3) || (sg.k & 3)) return false;
    }
    if (a.splits > 1 && (a.kper & 3)) return false;
    // the vector path addresses every operand by 32-bit byte offsets from its base (raw-buffer loads, 2 GiB window);
    // gathered segments (rowidx / rowkey) must fit that window too -- their extent is the caller's table
    size_t krows = 0;
    for (int s = 0; s < a.nseg; ++s) {
        const ASeg& sg = a.seg[s];
        if (sg.kw + sg.k > (int)krows) krows = (size_t)(sg.kw + sg.k);
        if (!sg.ptr || sg.k <= 0 || sg.rowidx || sg.rowkey) continue;
        const size_t rows = sg.rowmod > 0 ? (size_t)sg.rowmod : 256;    // plain segments are addressed per tile (<= 256 rows)
        if (rows * sg.ld * 4 >= (1ull << 31)) return false;
    }
    if ((bt ? (size_t)a.N : krows) * a.ldw * 4 >= (1ull << 31)) return false;
    return true;
}

int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Pick the configuration with the best estimated time on 256 CUs.  The launch is MFMA-bound when every CU holds
// at least two workgroups, and the workgroups of a CU share its matrix pipes, so time ~ (workgroups on the busiest
// CU) x (flops of one tile), times a penalty for thin tiles (operand bytes per flop) and for a lone 4-wave
// workgroup per CU (nothing to overlap its barriers with).  Calibrated on the bench shapes: 6400x12000x1000 picks
// 96x96 (105 vs 96 TFLOP/s for 128x128, whose 400 tiles leave half the CUs with one workgroup), the others 128x128.
int choose(const CfgEntry* t, int n, int M, int N, int splits = 1)
{
    int best = 0;
    double best_cost = 1e300;
    for (int i = 0; i < n; ++i) {
        if (t[i].fallback >= 0) continue;                  // LDS-DMA variants are selected by name (below), not by the cost model
        const long wgs = (long)ceil_div(M, t[i].BM) * ceil_div(N, t[i].CG) * (splits > 1 ? splits : 1);
        const long per_cu = (wgs + 255) / 256;
        const double tile = (double)t[i].BM * t[i].CG;
        const double traffic = 1.0 + 24.0 * (t[i].BM + t[i].CG) / tile;  // operand bytes per flop, relative
        const double lone = (per_cu == 1 && t[i].NT < 512) ? 1.4 : 1.0;
        const double cost = (double)per_cu * tile * traffic * lone;
        if (cost < best_cost) { best_cost = cost; best = i; }
    }
    return best;
}

// LSTM step: the gate-per-wave tile whose row count cuts M into ~4 row tiles, so that 4 x ceil(H/16) workgroups
// (~one per CU at H = 1000) cover the launch with one wave per SIMD each; measured on MI355X at H = 1000:
// M = 64: 21 us (64x16u: 39), M = 320: 98 us at K = 2500 (64x16u: 125), M = 384: 102 us (137).
int choose_lstm(int M)
{
    // beyond ~one workgroup per CU (M > 384: several row tiles per weight panel) 64- or 80-row tiles, whichever pads
    // M less, beat the 96-row ones by 10-30 % (measured at M = 448 ... 2304: the reference's default batch is 2048 / 2304 rows)
    if (M > 384) return ceil_div(M, 80) * 80 < ceil_div(M, 64) * 64 ? kLstmGw80L : kLstmGwFirst + 3;
    const int rows = ceil_div(ceil_div(M, 4), 16) * 16;       // 16, 32, ... rows per workgroup
    const int step = rows / 16;                                // 1..6 -> gw16 .. gw96
    if (dma_tiles()) {
        if (step == 4) return kLstmGw64D;                  // (tools/tune_dma.py: 16- and 32-row tiles are faster register-staged)
        if (step >= 6) return kLstmGw96D6;
    }
    if (step <= 1) return kLstmGw16k64;
    if (step == 5) return kLstmGw80L;
    if (step >= 6) return kLstmGw96L;
    return kLstmGwFirst + step - 1;
}

std::once_flag g_attr_once;
hipError_t g_attr_err = hipSuccess;        // first refusal of a dynamic-LDS request: reported by every launch instead of an opaque launch failure

// ---- launch profiler state
struct Pending { int cls, cfg; const char* name; double flops; hipEvent_t e0, e1; };
std::mutex g_prof_mu;
bool g_prof_on = false;
int g_prof_cls = -1, g_prof_cfg = -1;      // >= 0: only launches of this (class, tile cfg) are bracketed by events
std::vector<Pending> g_pending;
constexpr size_t kMaxPending = 1u << 16;    // launches between two prof_collect() calls that are kept (~100 bench steps)
std::vector<hipEvent_t> g_event_pool;

void set_lds_attrs()
{
    for (int epi : {(int)EPI_STORE, (int)EPI_LSTM, (int)EPI_PICK, (int)EPI_STORE_NT}) {
        int n;
        const CfgEntry* t = table(epi, &n);
        for (int i = 0; i < n; ++i) {
            for (KernelFn fn : {t[i].vec, t[i].scalar, t[i].vec_om, t[i].scalar_om}) {
                if (!fn) continue;
                const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                         t[i].lds_bytes);
                if (e != hipSuccess && g_attr_err == hipSuccess) g_attr_err = e;
            }
        }
    }
}

}  // namespace

void prof_enable(bool on)
{
    std::lock_guard<std::mutex> l(g_prof_mu);
    g_prof_on = on;
}
bool prof_on() { return g_prof_on; }
bool prof_wants(int cls, int cfg) { return g_prof_on && (g_prof_cls < 0 || (g_prof_cls == cls && (g_prof_cfg < 0 || g_prof_cfg == cfg))); }
void prof_filter(int cls, int cfg)
{
    std::lock_guard<std::mutex> l(g_prof_mu);
    g_prof_cls = cls;
    g_prof_cfg = cfg;
}

hipError_t prof_events(hipEvent_t* e0, hipEvent_t* e1)
{
    std::lock_guard<std::mutex> l(g_prof_mu);
    hipEvent_t* out[2] = {e0, e1};
    for (int i = 0; i < 2; ++i) {
        if (!g_event_pool.empty()) {
            *out[i] = g_event_pool.back();
            g_event_pool.pop_back();
        } else {
            hipError_t e = hipEventCreate(out[i]);
            if (e != hipSuccess) return e;
        }
    }
    return hipSuccess;
}

void prof_record(int cls, int cfg, const char* name, double flops, hipEvent_t e0, hipEvent_t e1)
{
    std::lock_guard<std::mutex> l(g_prof_mu);
    if (g_pending.size() >= kMaxPending) {             // nobody collects: stop recording rather than grow without bound
        g_event_pool.push_back(e0);
        g_event_pool.push_back(e1);
        return;
    }
    g_pending.push_back(Pending{cls, cfg, name, flops, e0, e1});
}

int prof_collect(ProfRow* rows, int max_rows)
{
    std::lock_guard<std::mutex> l(g_prof_mu);
    int n = 0;
    for (const Pending& p : g_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.e0, p.e1) != hipSuccess) ms = 0.f;
        int i = 0;
        for (; i < n; ++i)
            if (rows[i].cls == p.cls && rows[i].cfg == p.cfg) break;
        if (i == n) {
            if (n == max_rows) continue;
            rows[n++] = ProfRow{p.cls, p.cfg, 0, 0.0, 0.0, p.name};
        }
        rows[i].launches += 1;
        rows[i].ms += ms;
        rows[i].flops += p.flops;
        g_event_pool.push_back(p.e0);
        g_event_pool.push_back(p.e1);
    }
    g_pending.clear();
    return n;
}

int gemm_num_cfgs(int epi)
{
    int n;
    table(epi, &n);
    return n;
}

const char* gemm_cfg_name(int epi, int cfg)
{
    int n;
    const CfgEntry* t = table(epi, &n);
    return (cfg >= 0 && cfg < n) ? t[cfg].name : "?";
}

hipError_t launch_gemm(const GemmArgs& a, int epi, int cfg, hipStream_t st)
{
    std::call_once(g_attr_once, set_lds_attrs);
    if (g_attr_err != hipSuccess) return g_attr_err;
    int n;
    const CfgEntry* t = table(epi, &n);
    // vocab pick: 64x96 tiles put ~3 independent workgroups on every CU at M = (K+1)*B = 384 (750 tiles); measured
    // 109 us vs 131 us for one 96x192 8-wave workgroup per CU and 136 us for 64x128
    const bool auto_cfg = cfg < 0 || cfg >= n;
    if (auto_cfg) cfg = epi == EPI_LSTM ? choose_lstm(a.M) : (epi == EPI_PICK && a.M >= 32 ? choose_pick(a.M) : choose(t, n, a.M, a.N, a.splits));
    // many-tile store shapes (>= two rounds of workgroups): the LDS-DMA ring twin of the chosen tile -- 119-127 -> 130-135 TFLOP/s on the
    // step's batched products (tools/tune_store.py, round 6); short launches keep the register-staged tiles (their ring fills faster)
    // (in the step, profiles/r06_dma_ab.json: logits 128x128 122.7 -> 129.6, hoisted 96x96 117.7 -> 124.9, dX2 nt128x128 126.8 -> 133.5;
    //  dO2's nt96x96 -- K = 12000, W^T rows 48 KB apart -- went 124.1 -> 118.0 and keeps its register-staged tile)
    if (auto_cfg && dma_tiles() && (epi == EPI_STORE || epi == EPI_STORE_NT) && a.splits <= 1 && (cfg == 2 || (cfg == 6 && epi == EPI_STORE)) &&
        (long)ceil_div(a.M, t[cfg].BM) * ceil_div(a.N, t[cfg].CG) >= 512)
        cfg = cfg == 2 ? 8 : 9;
    if (t[cfg].fallback >= 0 && !can_vec(a, epi == EPI_STORE_NT)) cfg = t[cfg].fallback;       // LDS-DMA tiles: aligned operands only
    const CfgEntry& e = t[cfg];
    if (a.M <= 0 || a.N <= 0) return hipSuccess;
    const int mt = ceil_div(a.M, e.BM), nt = ceil_div(a.N, e.CG);
    GemmArgs a2 = a;
    a2.xcd_map = (mt <= 16 && nt >= 8) ? 1 : 0;
    // L2-sized supertiles for the many-tile store shapes (forward products and W^T data gradients): S2VT_SUP="gm,gn"
    // forces a shape (dev knob; "0" = the previous banded order)
    if (!a2.xcd_map && (epi == EPI_STORE || epi == EPI_STORE_NT) && a.splits <= 1 && (long)mt * nt >= 512 && mt >= 4) {
        static const int knob = [] {
            const char* e = getenv("S2VT_SUP");
            if (!e) return -1;
            int gm = 0, gn = 0;
            if (sscanf(e, "%d,%d", &gm, &gn) == 2 && gm > 0 && gn > 0) return gm * 1000 + gn;
            return 0;
        }();
        int gm = 0, gn = 0;
        if (knob > 0) { gm = knob / 1000; gn = knob % 1000; }
        else if (knob < 0) {
            // ~64-96 workgroups per XCD at a time (32 CUs x 2-3): 6 row tiles x 11 column tiles (whole rows of a grid with few
            // column tiles).  Measured (tools/ab_supertile_traffic.sh, profiles/r03_supertile_ab.jsonl): fetches of the logits
            // product 2035 -> 768 MB per launch, of dO2 3950 -> 1770; 8x8 / 4x8 / 4x16 within 4 % of it, 16x4 and 12x11
            // 25-50 % worse (a strip of 6 x 128 rows of A stays in the XCD's 4 MB L2 while the column tiles walk past it)
            gn = nt <= 12 ? nt : 11;
            gm = 6;
        }
        if (gm > 0 && gn > 0) {
            if (gn > nt) gn = nt;
            if (gm > mt) gm = mt;
            a2.xcd_map = 2; a2.sup_gm = gm; a2.sup_gn = gn;
        }
    }
    const unsigned gx = a2.xcd_map == 1 ? (unsigned)(mt * ceil_div(nt, 8) * 8) : (unsigned)(mt * nt);
    const dim3 grid(gx, (unsigned)(a.splits > 1 ? a.splits : 1), 1);
    KernelFn fn = can_vec(a, epi == EPI_STORE_NT) ? e.vec : e.scalar;
    if (a.omap || a.m_dev) {                                           // live-row launch: its own instantiations
        if (!e.vec_om) return hipErrorInvalidValue;
        fn = fn == e.vec ? e.vec_om : e.scalar_om;
    }
    const int pcls = epi;                                              // profiler class (0 store, 1 LSTM, 2 pick, 3 = TN kernel, 4 store with W^T)
    if (!prof_wants(pcls, cfg)) {
        hipLaunchKernelGGL(fn, grid, dim3(e.NT), e.lds_bytes, st, a2);
        return hipGetLastError();
    }
    hipEvent_t e0, e1;
    hipError_t pe = prof_events(&e0, &e1);
    if (pe != hipSuccess) return pe;
    double ksum = 0.0;
    for (int s = 0; s < a.nseg; ++s)
        if (a.seg[s].ptr && a.seg[s].k > 0) ksum += a.seg[s].k;
    const double cols = (double)a.N * (epi == EPI_LSTM ? 4 : 1);
    (void)hipEventRecord(e0, st);
    hipLaunchKernelGGL(fn, grid, dim3(e.NT), e.lds_bytes, st, a2);
    (void)hipEventRecord(e1, st);
    prof_record(pcls, cfg, e.name, 2.0 * a.M * ksum * cols, e0, e1);
    return hipGetLastError();
}

}  // namespace s2vt

#ifdef S2VT_STAMP
// dev build only: read and reset the per-segment clock sums of gemm_kernel (tools/stamp_loop.py)
extern "C" int s2vt_stamp_read(unsigned long long* out16)
{
    if (hipDeviceSynchronize() != hipSuccess) return -4;
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(s2vt::s2vt_stamp_acc), 16 * sizeof(unsigned long long)) != hipSuccess) return -4;
    unsigned long long z[16] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(s2vt::s2vt_stamp_acc), z, sizeof(z)) == hipSuccess ? 0 : -4;
}
#endif

