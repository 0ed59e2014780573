// stream_probe.hip -- how fast can a CU pull bytes from L2 / Infinity Cache into LDS (or registers), as a function of
// the bytes it keeps in flight?  Diagnostic only (nothing in the product includes it).
//
// Every forward contraction of the sampler (vocabulary pick, fused cell step; M = 64 and M = 384) moves ~20 GB/s per CU
// whatever its tile (profiles/NOTES.md, round 6): this probe measures the ceiling of that path with the MFMA work taken
// away.  One workgroup = NL loader waves; each wave keeps DEPTH 1-KiB pieces (64 lanes x 16 B) in flight:
//   mode dma : buffer_load_dwordx4 ... lds (global -> LDS, no registers), s_waitcnt vmcnt(DEPTH - 1) per piece
//   mode reg : buffer_load_dwordx4 to a DEPTH-deep register ring
// Patterns (what the product's operands look like):
//   private : workgroup b streams its own contiguous slice (a pre-packed weight slice: W2' of decode_loop.hip)
//   panel   : rows of ROWB bytes, row stride LDW bytes (a [k][n] weight panel as it lies in embed_word_W: 384-B row pieces,
//             48,000 B apart); SHARE consecutive same-XCD workgroups read the SAME panel (the row tiles of one column tile)
//   shared  : every workgroup reads the same buffer (the A operand / the state image)
//   hipcc --offload-arch=gfx950 -O3 -o stream_probe tools/micro/stream_probe.hip && ./stream_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;

struct Args {
    const float* src;
    unsigned long long wg_base_stride;   // bytes between the bases of consecutive "owners"
    int share;                           // consecutive same-XCD workgroups that read the same base (1: private)
    int rowb, ldw;                       // panel pattern: row piece bytes / row stride bytes (rowb == 0: contiguous)
    int gates, gstride;                  // > 1: a row is `gates` pieces of rowb bytes, gstride bytes apart (the four gate column groups of an LSTM weight)
    int pieces;                          // 1-KiB pieces per wave
    float* sink;
};

template <int NL, int DEPTH, bool DMA, int AUX = 0>
__global__ __launch_bounds__(64 * NL) void stream_kernel(const Args g)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int owner = (slot / g.share) * 8 + xcd;
    const float* base = g.src + (size_t)owner * (g.wg_base_stride / 4);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000);
    f32x4 ring[DMA ? 1 : DEPTH];
    float acc = 0.f;
    auto offset = [&](int p) {
        const unsigned lin = (unsigned)(p * NL + wave) * 1024u + (unsigned)lane * 16u;
        if (g.rowb == 0) return lin;
        const unsigned pc = lin / (unsigned)g.rowb, in = lin % (unsigned)g.rowb;
        if (g.gates > 1) return (pc / (unsigned)g.gates) * (unsigned)g.ldw + (pc % (unsigned)g.gates) * (unsigned)g.gstride + in;
        return pc * (unsigned)g.ldw + in;
    };
    auto issue = [&](int p, int s) __attribute__((always_inline)) {
        if constexpr (DMA) {
#if defined(__HIP_DEVICE_COMPILE__)      // (the host pass drops a kernel instantiation that names a builtin it does not know)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(smem + (s * NL + wave) * 256), 16, offset(p), 0, 0, AUX);
#endif
        } else {
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(ring[s]) : "v"(offset(p)), "s"(rs) : "memory");
        }
    };
#pragma unroll
    for (int s = 0; s < DEPTH - 1; ++s) issue(s, s);
    for (int p0 = 0; p0 < g.pieces; p0 += DEPTH) {
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) {
            const int p = p0 + s;
            if (p + DEPTH - 1 < g.pieces) issue(p + DEPTH - 1, (s + DEPTH - 1) % DEPTH);
            else asm volatile("s_nop 0");
            // the oldest piece has landed when at most DEPTH - 1 younger ones are outstanding (near the end fewer are: wait for all)
            if (p + DEPTH - 1 < g.pieces) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (!DMA) { asm volatile("" : "+v"(ring[s])); acc += ring[s][0]; }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (DMA) acc = smem[threadIdx.x];
    if (acc == 123.456f) g.sink[blockIdx.x] = acc;
}

// dword LDS-DMA: 64 lanes x 4 B = 256 B per wave-instruction (what a k%4-plane A image would need: 4-byte scatter granularity)
template <int NL, int DEPTH>
__global__ __launch_bounds__(64 * NL) void stream4_kernel(const Args g)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int owner = (slot / g.share) * 8 + xcd;
    const float* base = g.src + (size_t)owner * (g.wg_base_stride / 4);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000);
    auto issue = [&](int p, int s) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
        // a strided gather like the plane image's: lane l takes float (l % 8) * 4 of row l / 8 of a 32-float-wide block
        const unsigned lin = (unsigned)(p * NL + wave) * 1024u + (unsigned)(lane >> 3) * 128u + (unsigned)(lane & 7) * 16u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(smem + (s * NL + wave) * 64), 4, lin, 0, 0, 0);
#endif
    };
    const int pieces = g.pieces * 4;
#pragma unroll
    for (int s = 0; s < DEPTH - 1; ++s) issue(s, s);
    for (int p0 = 0; p0 < pieces; p0 += DEPTH) {
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) {
            const int p = p0 + s;
            if (p + DEPTH - 1 < pieces) { issue(p + DEPTH - 1, (s + DEPTH - 1) % DEPTH); asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (smem[threadIdx.x & 63] == 123.456f) g.sink[blockIdx.x] = 1.f;
}

template <int NL, int DEPTH>
void report4(const char* pattern, const Args& a0, size_t bytes_per_wg)
{
    Args a = a0;
    a.pieces = (int)(bytes_per_wg / 1024 / NL);
    const size_t lds = (size_t)DEPTH * NL * 256;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((stream4_kernel<NL, DEPTH>), dim3(256), dim3(64 * NL), lds, 0, a);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((stream4_kernel<NL, DEPTH>), dim3(256), dim3(64 * NL), lds, 0, a);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms / 20 * 1e3;
    // every instruction moves 256 B into LDS (and touches 8 x 16 B of each of 8 128-byte lines)
    const double per_cu = (double)a.pieces * 4 * NL * 256 / (us * 1e-6) / 1e9;
    printf("{\"pattern\": \"%s\", \"mode\": \"dma dword (256 B per instruction)\", \"loader_waves\": %d, \"depth_instr_per_wave\": %d, \"us\": %.1f, "
           "\"gb_s_per_cu\": %.1f, \"instr_per_us_per_wave\": %.2f}\n", pattern, NL, DEPTH, us, per_cu, (double)a.pieces * 4 / us);
    fflush(stdout);
}

// the same DMA stream with sc1 loads (aux 16: the load form of a cross-XCD hand-off, what the persistent kernels read their state image with)
template <int NL, int DEPTH>
void report_sc1(const char* pattern, const Args& a0, size_t bytes_per_wg)
{
    Args a = a0;
    a.pieces = (int)(bytes_per_wg / 1024 / NL);
    const size_t lds = (size_t)DEPTH * NL * 1024;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(stream_kernel<NL, DEPTH, true, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((stream_kernel<NL, DEPTH, true, 16>), dim3(256), dim3(64 * NL), lds, 0, a);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((stream_kernel<NL, DEPTH, true, 16>), dim3(256), dim3(64 * NL), lds, 0, a);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms / 20 * 1e3;
    printf("{\"pattern\": \"%s\", \"mode\": \"dma sc1\", \"loader_waves\": %d, \"depth_kib_per_wave\": %d, \"us\": %.1f, \"gb_s_per_cu\": %.1f}\n", pattern, NL, DEPTH, us,
           (double)a.pieces * NL * 1024 / (us * 1e-6) / 1e9);
    fflush(stdout);
}

template <int NL, int DEPTH, bool DMA>
double run(const Args& a, int grid, int reps)
{
    const size_t lds = DMA ? (size_t)DEPTH * NL * 1024 : 256;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(stream_kernel<NL, DEPTH, DMA>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((stream_kernel<NL, DEPTH, DMA>), dim3(grid), dim3(64 * NL), lds, 0, a);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((stream_kernel<NL, DEPTH, DMA>), dim3(grid), dim3(64 * NL), lds, 0, a);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps * 1e3;       // us per launch
}

template <int NL, int DEPTH, bool DMA>
void report(const char* pattern, const Args& a0, int wgs_per_cu, size_t bytes_per_wg)
{
    Args a = a0;
    a.pieces = (int)(bytes_per_wg / 1024 / NL);
    const int grid = 256 * wgs_per_cu;
    const double us = run<NL, DEPTH, DMA>(a, grid, 20);
    const double per_cu = (double)a.pieces * NL * 1024 * wgs_per_cu / (us * 1e-6) / 1e9;
    printf("{\"pattern\": \"%s\", \"mode\": \"%s\", \"loader_waves\": %d, \"depth_kib_per_wave\": %d, \"wgs_per_cu\": %d, \"in_flight_kib_per_cu\": %d, "
           "\"kib_per_wg\": %zu, \"us\": %.1f, \"gb_s_per_cu\": %.1f, \"tb_s_chip\": %.2f}\n",
           pattern, DMA ? "dma" : "reg", NL, DEPTH, wgs_per_cu, DEPTH * NL * wgs_per_cu, (size_t)a.pieces * NL, us, per_cu, per_cu * 256 / 1e3);
    fflush(stdout);
}

int main()
{
    const size_t total = 512ull << 20;
    float* buf;
    float* sink;
    CHECK(hipMalloc(&buf, total));
    CHECK(hipMalloc(&sink, 1 << 20));
    CHECK(hipMemset(buf, 0, total));
    Args priv{buf, 384 * 1024, 1, 0, 0, 0, 0, 0, sink};                  // 256 (x wgs) private 384-KiB slices: 96 MB per launch per WG-per-CU
    Args panel{buf + (32 << 20), 384, 6, 384, 48000, 0, 0, 0, sink};     // [1000][12000] fp32 matrix: 125 panels of 96 columns, 6 row tiles share one
    Args shared{buf + (48 << 20), 0, 1, 0, 0, 0, 0, 0, sink};           // everybody reads the same 1.5 MB
    Args xshared{buf + (64 << 20), 3 << 20, 1 << 20, 0, 0, 0, 0, 0, sink};   // one 3-MiB buffer per XCD (share = everything on the XCD): L2-resident
#define SWEEP(PAT, NAME, WGS, BYTES)                                              \
    report<1, 8, true>(NAME, PAT, WGS, BYTES); report<1, 16, true>(NAME, PAT, WGS, BYTES); report<1, 32, true>(NAME, PAT, WGS, BYTES);   \
    report<2, 16, true>(NAME, PAT, WGS, BYTES); report<4, 8, true>(NAME, PAT, WGS, BYTES); report<4, 16, true>(NAME, PAT, WGS, BYTES);   \
    report<4, 32, true>(NAME, PAT, WGS, BYTES); report<8, 16, true>(NAME, PAT, WGS, BYTES); report<8, 4, true>(NAME, PAT, WGS, BYTES);   \
    report<12, 8, true>(NAME, PAT, WGS, BYTES); report<16, 8, true>(NAME, PAT, WGS, BYTES); report<16, 4, true>(NAME, PAT, WGS, BYTES);  \
    report<4, 4, false>(NAME, PAT, WGS, BYTES); report<4, 8, false>(NAME, PAT, WGS, BYTES); report<4, 16, false>(NAME, PAT, WGS, BYTES); \
    report<8, 16, false>(NAME, PAT, WGS, BYTES);
    SWEEP(priv, "private contiguous 384 KiB per WG (96 MB per launch: Infinity Cache / HBM)", 1, 384 * 1024)
    {
        const char* nm = "panel rows of 384 B, stride 48000 B, 6 WGs per panel on one XCD (embed_word_W as the pick reads it), 3 WGs per CU";
        report<1, 8, true>(nm, panel, 3, 384 * 1024); report<1, 16, true>(nm, panel, 3, 384 * 1024); report<1, 32, true>(nm, panel, 3, 384 * 1024);
        report<2, 16, true>(nm, panel, 3, 384 * 1024); report<4, 8, true>(nm, panel, 3, 384 * 1024); report<4, 12, true>(nm, panel, 3, 384 * 1024);
        report<4, 4, false>(nm, panel, 3, 384 * 1024); report<4, 8, false>(nm, panel, 3, 384 * 1024); report<4, 16, false>(nm, panel, 3, 384 * 1024);
    }
    SWEEP(shared, "one 1.5 MB buffer read by every WG (the A operand at 384 rows)", 1, 1536 * 1024)
    SWEEP(xshared, "one 3 MiB buffer per XCD read by its 32 CUs (L2-resident)", 1, 3 << 20)
    {
        // the M = 64 sampler step's weight operands as they lie in memory, one workgroup per CU, every workgroup walking k in step:
        //   pick 32x96: [1000][12000] fp32, workgroup = rows of 384 B 48,000 B apart, 2 row tiles share a panel      (48 MB distinct per launch)
        //   cell gw16 : [1500][4000] fp32 (the emb + h2 rows of lstm2_W), workgroup = 4 x 64 B per row (one per gate), 4 row tiles share   (24 MB)
        // against the same bytes as private contiguous slices (what a per-workgroup packed copy would be)
        Args pick64{buf + (32 << 20), 384, 2, 384, 48000, 0, 0, 0, sink};
        Args cell64{buf + (96 << 20), 64, 4, 64, 16000, 4, 4000, 0, sink};
        Args privc{buf, 96 * 1024 * 4, 4, 0, 0, 0, 0, 0, sink};      // 4 workgroups share a contiguous 384-KiB slice (packed W2 slice of a unit group)
        Args privp{buf, 384 * 1024, 2, 0, 0, 0, 0, 0, sink};         // 2 workgroups share a contiguous 384-KiB panel (packed embed_word_W panel)
        report<4, 8, true>("pick M=64 operand: [1000][12000] panel of 96 columns, 2 row tiles share", pick64, 1, 384 * 1024);
        report<8, 8, true>("pick M=64 operand: [1000][12000] panel of 96 columns, 2 row tiles share", pick64, 1, 384 * 1024);
        report<4, 8, true>("pick M=64 operand PACKED: contiguous 384 KiB per panel, 2 share", privp, 1, 384 * 1024);
        report<8, 8, true>("pick M=64 operand PACKED: contiguous 384 KiB per panel, 2 share", privp, 1, 384 * 1024);
        report<4, 8, true>("cell M=64 operand: [1500][4000], 4 x 64 B per row, 4 row tiles share", cell64, 1, 384 * 1024);
        report<8, 8, true>("cell M=64 operand: [1500][4000], 4 x 64 B per row, 4 row tiles share", cell64, 1, 384 * 1024);
        report<4, 8, true>("cell M=64 operand PACKED: contiguous 384 KiB per unit group, 4 share", privc, 1, 384 * 1024);
        report<8, 8, true>("cell M=64 operand PACKED: contiguous 384 KiB per unit group, 4 share", privc, 1, 384 * 1024);
    }
    report_sc1<1, 16>("one 1.5 MB buffer read by every WG, sc1 loads", shared, 1536 * 1024);
    report_sc1<4, 16>("one 1.5 MB buffer read by every WG, sc1 loads", shared, 1536 * 1024);
    report_sc1<8, 16>("one 1.5 MB buffer read by every WG, sc1 loads", shared, 1536 * 1024);
    report<4, 16, true>("one 1.5 MB buffer read by every WG, plain loads (same run)", shared, 1, 1536 * 1024);
    report4<1, 16>("one 3 MiB buffer per XCD, strided 4-byte gather", xshared, 768 * 1024);
    report4<4, 16>("one 3 MiB buffer per XCD, strided 4-byte gather", xshared, 768 * 1024);
    report4<4, 32>("one 3 MiB buffer per XCD, strided 4-byte gather", xshared, 768 * 1024);
    report4<8, 16>("one 3 MiB buffer per XCD, strided 4-byte gather", xshared, 768 * 1024);
    report4<16, 16>("one 3 MiB buffer per XCD, strided 4-byte gather", xshared, 768 * 1024);
    return 0;
}
