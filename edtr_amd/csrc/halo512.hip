// Halo tile, 512-pixel unit (tile 17 of edtr_igemm, round 5): 3x3 / stride 1 / pad 1 convolutions whose output image is a
// multiple of 32 x 16 pixels and whose N is a multiple of 128 — the VAE ResnetBlock convolutions (reference model/vae.py:74-124,
// 479-525) at the 512^2 ... 64^2 levels.
//
// Why a second geometry.  The 256-pixel halo tile (igemm.hip, tile 16) gives each of its 8 waves 128 pixels x 64 channels x ONE
// HALF of K, so the two halves must meet in an fp32 staging tile in LDS: in-kernel stamps of a K = 1152 unit read 2.3k cycles of
// address set-up + 3.1k of first-load flight + 22.7k of multiply loop + 10.9k of epilogue (accumulators -> LDS, second half added
// in LDS, row loop out of LDS): 42 % of a unit's life is not the loop, and with one 144-KiB workgroup per CU nothing else runs
// under it.  Here the unit is 32 x 16 = 512 output pixels x 128 channels and a wave owns 128 pixels x 64 channels over ALL of K
// (the same 128 accumulator registers), so
//   * a weight slice is multiplied against twice the pixels (L2 -> LDS weight traffic per FLOP halves),
//   * the fixed costs (set-up, first-load flight) are paid once per 512 pixels,
//   * the accumulators are FINAL in registers: the epilogue needs no LDS at all.  The weight rows are fed to the MFMA in a
//     permuted order (the permutation is the DMA's source address: free) such that a lane ends up with two runs of 8 CONSECUTIVE
//     output channels of its pixel: bias / time-embedding row / residual / 16-byte stores straight from registers, four lanes
//     writing 64 contiguous bytes of a pixel row.
// LDS: the input patch is staged per 32-CHANNEL chunk (34 x 18 pixels x 64 B = 38.25 KiB, double buffered) — the 16x16x32 MFMA
// takes exactly one chunk per instruction — plus a ring of three 8-KiB weight slices (one per tap): 106 KiB.  Pixel (py, px) sits
// at (34 py + px) * 64 B with its four 16-byte channel groups XOR-ed by kKey[px & 7]; that table was found by exhaustive search
// against the actual lane groups of ds_read_b128 ({0-3, 12-15, 20-27}, ... MI355X_MICROARCH.md, LDS) so that the A fragments
// of all three dx and the B fragments are conflict-free with 64-byte pixels (no (px & 3)-style closed form is).
// Same eight-wave ping-pong as tile 16: waves w / w + 4 share a SIMD and run half a phase (16 MFMAs) apart, counted vmcnt waits,
// raw s_barriers, one weight piece per wave and tap, five patch pieces per wave and chunk.
// a_gn (GroupNorm + SiLU of the input applied in the patch staging) as in tile 16; per output pixel it normalises 1.20 patch
// pixels instead of 1.27.
// The file also holds tile 20 (igemm_halo160_kernel, below): the same ideas on 16 x 16 pixels x 160 channels for N % 160 == 0.
#include <stdlib.h>
#include <type_traits>
#include "common.h"

// Diagnostic build only (tools/exp/halo512_stamps.py compiles with -DEDTR_STAMPS): thread 0 of every workgroup stores s_memtime at
// phase boundaries into p.workspace (unused by this tile otherwise).  The product library contains no stamp code.
#ifdef EDTR_STAMPS
#define H5_STAMP(i)                                                                                     \
    do {                                                                                                \
        if (threadIdx.x == 0 && p.workspace) {                                                          \
            uint64_t* sb__ = static_cast<uint64_t*>(p.workspace) + (size_t)blockIdx.x * 16;             \
            if ((i) == 14) sb__[7] = (uint64_t)__builtin_amdgcn_s_getreg(63492) | ((uint64_t)__builtin_amdgcn_s_getreg(63508) << 32); \
            sb__[i] = ((i) >= 14) ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime();    \
        }                                                                                               \
    } while (0)
#else
#define H5_STAMP(i)
#endif

namespace {

constexpr uint32_t kOob = 0xFFFFFF00u;     // >= num_records - 15: the buffer range check turns the lane's 16 bytes into zeros

__device__ __forceinline__ u32x4 srd_of(const void* base) {
    const uint64_t b = reinterpret_cast<uint64_t>(base);
    u32x4 srd;
    srd.x = __builtin_amdgcn_readfirstlane((uint32_t)b);
    srd.y = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32) & 0xffffu);   // stride 0 (raw buffer)
    srd.z = 0xFFFFFF00u;
    srd.w = 0x00020000u;
    return srd;
}

// buffer-addressed LDS-DMA: lane i's 16 bytes (SRD base + voff + soff) land at LDS byte lds_addr + 16 i
__device__ __forceinline__ void dma(uint32_t voff, const u32x4& srd, uint32_t soff, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds"
                 :
                 : "v"(voff), "s"(srd), "s"(lds_addr), "s"(soff)
                 : "memory");
}

template <int V> using IC = std::integral_constant<int, V>;

constexpr int CK = 32;                          // channels per chunk = K of one 16x16x32 MFMA
constexpr int WTAP = 128 * CK * 2;              // 8 KiB: one tap's [128][32] weight slice
namespace h1 {                                  // tile 17
constexpr int PW = 34, PROWS = 18, PPIX = PW * PROWS, PROWB = PW * 64;
constexpr int NPP = 5;                          // patch pieces (1 KiB = 16 pixels) per wave and chunk: 40 >= 38.25
constexpr int PATCHB = 40 * 1024;
constexpr int W_BASE = 2 * PATCHB;
constexpr int TBL = W_BASE + 3 * WTAP;          // two 1-KiB slots for the (scale, shift) rows of a chunk (a_gn)
constexpr int RED = 128 * 1024;                 // epilogue: [0, 128 KiB) = the unit's 16-bit residual (16 KiB per wave), then the GroupNorm partial sums
constexpr int LDS_BYTES = RED + 4096;
}

// 2-bit XOR key of a 64-byte LDS row (a patch pixel or a weight row), by row index mod 8: {3, 3, 0, 1, 0, 1, 3, 2}
__device__ __forceinline__ int key8(int x) {
    constexpr uint32_t kKey = 3u | (3u << 2) | (0u << 4) | (1u << 6) | (0u << 8) | (1u << 10) | (3u << 12) | (2u << 14);
    return (int)((kKey >> (2 * (x & 7))) & 3u);
}

// sum over the 16 lanes of a DPP row (lanes that share lane >> 4), result in every lane: four v_add_f32 with DPP operands
// (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror) instead of four ds_bpermute round trips
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
}

template <int N> __device__ __forceinline__ void wait_vm() {
    if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else static_assert(N >= 1 && N <= 4, "wait_vm");
}

// ---- epilogue of the halo kernels of this file, straight from the accumulators (no LDS staging).  NW = waves per workgroup
// (8: tile 17, wave = 4 rows x 32 pixels; 4 would be 8 rows x 16 pixels: the removed two-workgroups-per-CU experiment), mu0 = the wave's first pixel, RED_OFF = LDS offset of
// the GroupNorm scratch (below it: 16 KiB per wave for the residual)
template <typename T, int NW, int RED_OFF>
__device__ __forceinline__ void direct_epilogue(const edtr_igemm_params& p, f32x4 (&acc)[8][4], char* smem, uint32_t smem_base, int lane, int wave,
                                                int wc, int64_t mu0, int n0, int img, int tm) {
    auto blk_off = [&](int mb) { return NW == 8 ? (mb & 3) * p.OW + 16 * (mb >> 2) : mb * p.OW; };     // pixel block mb of the wave, from mu0
    //   acc[mb][nb][i] = pixel (row 4 q + (mb & 3), column 16 (mb >> 2) + l15),
    // channel n0 + wc 64 + 32 (nb >> 1) + 8 lq + 4 (nb & 1) + i: per half hp = nb >> 1 a lane holds EIGHT consecutive channels of its
    // pixel.  One pass per hp (the pass's 64 accumulator registers die as it goes; per-channel addends and GroupNorm sums are
    // 8 + 16 registers instead of twice that), four pixel blocks at a time with their residual vectors requested together.
    // (lane-derived epilogue values come from an OPAQUE copy of the lane index: otherwise the compiler computes them at kernel entry
    //  and carries them — spilled — through the main loop, whose register file is full)
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int tid = (int)threadIdx.x;
    const int l15e = lane_e & 15, lqe = lane_e >> 4;
    const float alpha = p.alpha;
    const bool gn_acc = p.gn_partial != nullptr;
    // addresses = a wave-uniform 64-bit base per pixel block (scalar registers) + ONE 32-bit lane offset per tensor: with 64-bit
    // per-lane addresses the compiler keeps the 8 blocks' pointers alive across the two passes and spills
    const int nu = n0 + wc * 64;                         // + 32 hp
    const uint32_t lo_out = (uint32_t)(l15e * p.ldc + 8 * lqe), lo_res = (uint32_t)(l15e * p.ldr + 8 * lqe), lo_16 = (uint32_t)(l15e * p.ld16 + 8 * lqe);
    float* red = reinterpret_cast<float*>(smem + RED_OFF);   // [8 waves][64 channels][2]: the main loop's buffers are dead (every wave is
                                                         // past its last fragment read and its last DMA has landed)

    // 16-bit output: WHOLE 128-byte lines per store / load instruction.  A lane's two 16-byte runs (hp = 0 / 1) lie 64 bytes apart in
    // its pixel's row, and the four lanes of a pixel cover 64 contiguous bytes per instruction: stored as they sit, an instruction
    // touches 16 pixel rows with half a line each, and the vector-memory path runs at its LINE rate, not its byte rate (stamps:
    // ~64 cycles per wave-instruction whatever it moves: 16 B/clk/CU, a 16.5k-cycle epilogue with a residual).  So lanes (pixel P,
    // lq) and (pixel P + 8, lq) — eight lanes apart in their DPP row — swap one run each (row_ror:8): instruction 1 then writes
    // both runs of pixels 0..7 (lanes < 8: own run 0; lanes >= 8: the partner's run 1), instruction 2 those of pixels 8..15: eight
    // whole lines per instruction.  The residual is fetched in the same shape and swapped back before the add, so the
    // arithmetic and its single rounding are unchanged.
    auto ror8 = [&](const U4& v) {
        U4 r;
        r.x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.x, 0x128, 0xF, 0xF, false);
        r.y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.y, 0x128, 0xF, 0xF, false);
        r.z = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.z, 0x128, 0xF, 0xF, false);
        r.w = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.w, 0x128, 0xF, 0xF, false);
        return r;
    };
    auto sel = [&](bool c, const U4& a, const U4& b) {
        U4 r;
        r.x = c ? a.x : b.x; r.y = c ? a.y : b.y; r.z = c ? a.z : b.z; r.w = c ? a.w : b.w;
        return r;
    };
    auto finish16 = [&](auto res_c) {
        constexpr bool RES = decltype(res_c)::value;
        const bool lo8 = l15e < 8;
        const int pa = l15e & 7, co = (lo8 ? 0 : 32) + 8 * lqe;       // instruction 1: pixel pa, instruction 2: pixel pa + 8; channel offset of this lane's 16 bytes
        const uint32_t lo_o1 = (uint32_t)(pa * p.ldc + co), lo_o2 = lo_o1 + 8u * (uint32_t)p.ldc;
        const uint32_t lo_r1 = (uint32_t)(pa * p.ldr + co), lo_r2 = lo_r1 + 8u * (uint32_t)p.ldr;
        float cb[2][8], gs[2][8], gq[2][8];
#pragma unroll
        for (int hp = 0; hp < 2; ++hp) {
            const int nl = nu + 32 * hp + 8 * lqe;
#pragma unroll
            for (int j = 0; j < 8; ++j) { cb[hp][j] = 0.0f; gs[hp][j] = 0.0f; gq[hp][j] = 0.0f; }
            if (p.bias_n) {
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias_n + nl), b1 = *reinterpret_cast<const f32x4*>(p.bias_n + nl + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { cb[hp][j] = b0[j]; cb[hp][j + 4] = b1[j]; }
            }
            if (p.rowvec) {
                const float* rv = p.rowvec + (int64_t)img * p.rowvec_ld + nl;
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(rv), b1 = *reinterpret_cast<const f32x4*>(rv + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { cb[hp][j] += b0[j]; cb[hp][j + 4] += b1[j]; }
            }
        }
        // The residual goes global -> LDS by DMA, all 16 KiB of the wave at once (the main loop's LDS is dead; registers could hold
        // two blocks ahead at most, and eight dependent round trips to HBM cost 10k cycles), in the whole-line shape above: piece
        // 2 mb = instruction 1's lanes, piece 2 mb + 1 = instruction 2's; a lane then reads both of ITS runs from one piece
        // (lanes < 8 of a row: piece 2 mb at lane and lane + 8; lanes >= 8: piece 2 mb + 1 at lane - 8 and lane).  vmcnt is in-order
        // and two stores follow every block: 14 operations may stay in flight behind the block's two pieces, always.
        char* const rl = smem + wave * 16384;
        if constexpr (RES) {
            pin8(cb[0]); pin8(cb[1]);                   // (the bias loads are consumed BEFORE the DMAs: a compiler-placed wait on them would drain the DMAs too)
            const u32x4 srd_r = srd_of(p.residual);
#pragma unroll
            for (int mb = 0; mb < 8; ++mb) {
                const int64_t mu = mu0 + blk_off(mb);
                const uint32_t so = (uint32_t)((mu * p.ldr + nu) * 2);
                dma(lo_r1 * 2, srd_r, so, smem_base + wave * 16384 + (2 * mb) * 1024);
                dma(lo_r2 * 2, srd_r, so, smem_base + wave * 16384 + (2 * mb + 1) * 1024);
            }
        }
        const int rd0 = (lo8 ? 0 : 1024) + (lane_e & ~8) * 16, rd1 = (lo8 ? 0 : 1024) + (lane_e | 8) * 16;
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) {
            const int64_t mu = mu0 + blk_off(mb);
            float f[2][8];
#pragma unroll
            for (int hp = 0; hp < 2; ++hp)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f[hp][i] = __builtin_fmaf(acc[mb][2 * hp][i], alpha, cb[hp][i]);
                    f[hp][i + 4] = __builtin_fmaf(acc[mb][2 * hp + 1][i], alpha, cb[hp][i + 4]);
                }
            if constexpr (RES) {
                asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
                const U4 q0 = *reinterpret_cast<const U4*>(rl + mb * 2048 + rd0), q1 = *reinterpret_cast<const U4*>(rl + mb * 2048 + rd1);
                float rf[8];
                unpack8<T>(q0, rf);
#pragma unroll
                for (int i = 0; i < 8; ++i) f[0][i] += rf[i];
                unpack8<T>(q1, rf);
#pragma unroll
                for (int i = 0; i < 8; ++i) f[1][i] += rf[i];
            }
#pragma unroll
            for (int hp = 0; hp < 2; ++hp) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { gs[hp][i] += f[hp][i]; gq[hp][i] += f[hp][i] * f[hp][i]; }
                pin8(gs[hp]);        // (keeps the sums HERE: the optimizer otherwise sinks them into the `if (gn_acc)` below and carries
                pin8(gq[hp]);        //  every finished value there — through scratch)
            }
            const U4 h0 = pack8<T>(f[0]), h1 = pack8<T>(f[1]);
            const U4 got = ror8(sel(lo8, h1, h0));
            uint16_t* op = static_cast<uint16_t*>(p.out) + (mu * p.ldc + nu);
#ifdef H5_NOSTORE      // (diagnostic builds of tools/exp/halo512_stamps.py only)
            if (p.alpha == 12345.0f)
#endif
            {
            stg16(op + lo_o1, sel(lo8, h0, got));
            stg16(op + lo_o2, sel(lo8, got, h1));
            }
        }
        if (gn_acc) {
#pragma unroll
            for (int hp = 0; hp < 2; ++hp) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { gs[hp][j] = row16_sum(gs[hp][j]); gq[hp][j] = row16_sum(gq[hp][j]); }
                if (l15e == 0) {
                    float* dst = red + (wave * 64 + 32 * hp + 8 * lqe) * 2;
#pragma unroll
                    for (int j = 0; j < 8; ++j) { dst[2 * j] = gs[hp][j]; dst[2 * j + 1] = gq[hp][j]; }
                }
            }
        }
    };
    // (every launch-uniform option is a template parameter: tested inside the unrolled block loop, each one cuts the loop into
    // basic blocks that run at instruction latency with one wave per SIMD — DESIGN.md §4, shared epilogue)
    auto finish = [&](auto out32_c, auto res_c, auto mir_c) {
        constexpr bool OUT32 = decltype(out32_c)::value, MIR = decltype(mir_c)::value && OUT32;
        constexpr int RES = decltype(res_c)::value;      // 0 none, 1 16-bit, 2 fp32
        constexpr int GJ = 4;
#pragma unroll
        for (int hp = 0; hp < 2; ++hp) {
            const int nl = n0 + wc * 64 + 32 * hp + 8 * lqe;      // first of this lane's eight channels
            float cb[8], gs[8], gq[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { cb[j] = 0.0f; gs[j] = 0.0f; gq[j] = 0.0f; }
            if (p.bias_n) {
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias_n + nl), b1 = *reinterpret_cast<const f32x4*>(p.bias_n + nl + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { cb[j] = b0[j]; cb[j + 4] = b1[j]; }
            }
            if (p.rowvec) {                              // the time-embedding row of the unit's image (a unit never leaves its image)
                const float* rv = p.rowvec + (int64_t)img * p.rowvec_ld + nl;
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(rv), b1 = *reinterpret_cast<const f32x4*>(rv + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { cb[j] += b0[j]; cb[j + 4] += b1[j]; }
            }
#pragma unroll
            for (int grp = 0; grp < 8 / GJ; ++grp) {
                U4 r16[GJ];
                f32x4 r32[GJ][2];
                if constexpr (RES != 0) {
#pragma unroll
                    for (int j = 0; j < GJ; ++j) {
                        const int mb = grp * GJ + j;
                        const int64_t mu = mu0 + blk_off(mb);
                        if constexpr (RES == 1) {
                            r16[j] = ldg16(static_cast<const uint16_t*>(p.residual) + (mu * p.ldr + nu + 32 * hp) + lo_res);
                        } else {
                            const float* rp = static_cast<const float*>(p.residual) + (mu * p.ldr + nu + 32 * hp) + lo_res;
                            r32[j][0] = *reinterpret_cast<const f32x4*>(rp);
                            r32[j][1] = *reinterpret_cast<const f32x4*>(rp + 4);
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < GJ; ++j) {
                    const int mb = grp * GJ + j;
                    const int64_t mu = mu0 + blk_off(mb);
                    float f[8];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        f[i] = __builtin_fmaf(acc[mb][2 * hp][i], alpha, cb[i]);
                        f[i + 4] = __builtin_fmaf(acc[mb][2 * hp + 1][i], alpha, cb[i + 4]);
                    }
                    if constexpr (RES == 1) {
                        float rf[8];
                        unpack8<T>(r16[j], rf);
#pragma unroll
                        for (int i = 0; i < 8; ++i) f[i] += rf[i];
                    } else if constexpr (RES == 2) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) { f[i] += r32[j][0][i]; f[i + 4] += r32[j][1][i]; }
                    }
                    const int64_t oidx = mu * p.ldc + nu + 32 * hp;          // wave-uniform
                    if constexpr (OUT32) {
                        float* o = static_cast<float*>(p.out) + oidx + lo_out;
                        f32x4 o0, o1;
                        o0[0] = f[0]; o0[1] = f[1]; o0[2] = f[2]; o0[3] = f[3];
                        o1[0] = f[4]; o1[1] = f[5]; o1[2] = f[6]; o1[3] = f[7];
                        *reinterpret_cast<f32x4*>(o) = o0;
                        *reinterpret_cast<f32x4*>(o + 4) = o1;
                        if constexpr (MIR) stg16(static_cast<uint16_t*>(p.out16) + (mu * p.ld16 + nu + 32 * hp) + lo_16, pack8<T>(f));
                    } else {
                        stg16(static_cast<uint16_t*>(p.out) + oidx + lo_out, pack8<T>(f));
                    }
                    // (unconditional: nearly every launch of this tile wants the statistics, and a launch-uniform test here would cut
                    //  the unrolled loop into basic blocks)
#pragma unroll
                    for (int i = 0; i < 8; ++i) { gs[i] += f[i]; gq[i] += f[i] * f[i]; }
                    pin8(gs);        // (keeps the sums HERE: the optimizer otherwise sinks them into the `if (gn_acc)` below and carries
                    pin8(gq);        //  all 64 finished values of the pass there — through scratch)
                }
            }
            if (gn_acc) {        // fold the 16 pixel lanes that share this channel run; lane l15 == 0 publishes the wave's 128-pixel sums
#pragma unroll
                for (int j = 0; j < 8; ++j) { gs[j] = row16_sum(gs[j]); gq[j] = row16_sum(gq[j]); }
                if (l15e == 0) {
                    float* dst = red + (wave * 64 + 32 * hp + 8 * lqe) * 2;
#pragma unroll
                    for (int j = 0; j < 8; ++j) { dst[2 * j] = gs[j]; dst[2 * j + 1] = gq[j]; }
                }
            }
        }
    };
    {
        using std::true_type;
        using std::false_type;
        const bool res32 = p.residual && p.residual_f32, res16 = p.residual && !p.residual_f32, mir = p.out16 != nullptr;
        if (!p.out_f32) {
            if (res16) finish16(true_type{});
            else if (res32) finish(false_type{}, IC<2>{}, false_type{});
            else finish16(false_type{});
        } else if (mir) {
            if (res32) finish(true_type{}, IC<2>{}, true_type{});
            else if (res16) finish(true_type{}, IC<1>{}, true_type{});
            else finish(true_type{}, IC<0>{}, true_type{});
        } else {
            if (res32) finish(true_type{}, IC<2>{}, false_type{});
            else if (res16) finish(true_type{}, IC<1>{}, false_type{});
            else finish(true_type{}, IC<0>{}, false_type{});
        }
    }
    H5_STAMP(4);
    if (gn_acc) {
        // the four pixel quarters of a channel half meet in LDS: per-channel sum / sum of squares of the unit's 512 pixels
        __syncthreads();
        if (tid < 128) {
            const int wcc = tid >> 6, cl = tid & 63;
            float a = 0.0f, s = 0.0f;
#pragma unroll
            for (int k = 0; k < NW / 2; ++k) { a += red[((2 * k + wcc) * 64 + cl) * 2]; s += red[((2 * k + wcc) * 64 + cl) * 2 + 1]; }
            float* dst = p.gn_partial + ((int64_t)(NW / 2 * tm) * gn_ld_of(p) + n0 + tid) * 2;  // NW / 2 slots of 128 rows per unit: the first takes the sums
            dst[0] = a;
            dst[1] = s;
#pragma unroll
            for (int k = 1; k < NW / 2; ++k) { dst[2 * k * (int64_t)gn_ld_of(p)] = 0.0f; dst[2 * k * (int64_t)gn_ld_of(p) + 1] = 0.0f; }
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(512, 1) igemm_halo512_kernel(const edtr_igemm_params p) {
    using namespace h1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    H5_STAMP(0); H5_STAMP(14);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave >> 2, q = wave >> 1, wc = wave & 1;     // ping-pong group, pixel quarter (patch rows 4 q .. 4 q + 3), channel half
    const int l15 = lane & 15, lq = lane >> 4;

    const int tw = p.OW >> 5, tpi = tw * (p.OH >> 4);
    const int nbm = (p.M / (p.OH * p.OW)) * tpi, nbn = p.N >> 7;
    int bid = blockIdx.x;
    {       // an XCD (blockIdx mod 8) owns a contiguous range of units: the column tiles of a patch share its L2
        const int nblk = nbm * nbn, qq = nblk >> 3, r = nblk & 7, x = bid & 7, j = bid >> 3;
        bid = (x < r ? x * (qq + 1) : r * (qq + 1) + (x - r) * qq) + j;
    }
    const int tm = bid / nbn, tn = bid - tm * nbn;
    const int img = tm / tpi, tr = tm - img * tpi, ty = tr / tw, tx = tr - ty * tw;
    const int oy0 = ty * 16, ox0 = tx * 32, n0 = tn * 128;
    const int sy0 = oy0 - 1, sx0 = ox0 - 1;
    const int m0 = (img * p.OH + oy0) * p.OW + ox0;

    const uint16_t* a1 = static_cast<const uint16_t*>(p.a1);
    const uint32_t smem_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    const u32x4 srd_a = srd_of(a1);
    const u32x4 srd_w = srd_of(p.w);
    const int Cin = p.C1, nchunk = Cin / CK;

    // ---- staging geometry
    // weight piece = wave: LDS rows R = 16 wave .. + 15 of the slice, lane -> (R, slot).  LDS row R = wcR 64 + nb 16 + r holds the
    // weights of output channel wcR 64 + 32 (nb >> 1) + 8 (r >> 2) + 4 (nb & 1) + (r & 3): the MFMA's row block nb then leaves
    // lane (pixel, lq) with channels 32 (nb >> 1) + 8 lq + 4 (nb & 1) + (0..3) — two runs of eight consecutive channels
    uint32_t voff_w;
    {
        const int R = wave * 16 + (lane >> 2), slot = lane & 3;
        const int r = R & 15, nb = (R >> 4) & 3;
        const int n = n0 + (R & 64) + 32 * (nb >> 1) + 8 * (r >> 2) + 4 * (nb & 1) + (r & 3);
        const int c = slot ^ key8(R);
        voff_w = (uint32_t)(((int64_t)n * p.ldw + c * 8) * 2);
    }
    auto stage_w = [&](int chunk, int tap, int buf) {
        const uint32_t vo = chunk < nchunk ? voff_w : kOob;
        dma(vo, srd_w, (uint32_t)((tap * Cin + chunk * CK) * 2), smem_base + W_BASE + buf * WTAP + wave * 1024);
    };
    // the weight slices of taps 0 and 1 start their flight before the patch addresses are worked out
    stage_w(0, 0, 0);
    stage_w(0, 1, 1);
    uint32_t voff_p[NPP];
    uint32_t gn_bits = 0;          // per patch piece j: bits 4 j, 4 j + 1 = the 8-channel group of this lane's 16 bytes, bit 4 j + 3 = inside the image
#pragma unroll
    for (int j = 0; j < NPP; ++j) {
        const int u = (wave + 8 * j) * 64 + lane, pp = u >> 2, slot = u & 3;
        const int py = pp / PW, px = pp - py * PW;
        const int iy = sy0 + py, ix = sx0 + px;
        const bool ok = pp < PPIX && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
        const int c = slot ^ key8(px);
        voff_p[j] = ok ? (uint32_t)(((((int64_t)img * p.IH + iy) * p.IW + ix) * p.ld1 + c * 8) * 2) : kOob;
        if (ok) gn_bits |= (uint32_t)(8 | c) << (4 * j);
    }
    auto stage_p = [&](int chunk, int j, int par) {
        const uint32_t vo = chunk < nchunk ? voff_p[j] : kOob;
        dma(vo, srd_a, (uint32_t)(chunk * CK * 2), smem_base + par * PATCHB + (wave + 8 * j) * 1024);
    };
    const bool gnf = p.a_gn != nullptr;
    const u32x4 srd_t = srd_of(gnf ? p.a_gn + (int64_t)img * Cin * 2 : reinterpret_cast<const float*>(a1));
    auto stage_t = [&](int chunk, int par) {                 // (scale, shift) of the chunk's 32 channels: 256 bytes, lanes 0..15
        const uint32_t vo = (chunk < nchunk && lane < 16) ? (uint32_t)(lane * 16) : kOob;
        dma(vo, srd_t, (uint32_t)(chunk * CK * 8), smem_base + TBL + par * 1024);
    };
    auto gn_piece = [&](int j, int par) {
        const uint32_t bits = gn_bits >> (4 * j);
        if (bits & 8) {
            char* qp = smem + par * PATCHB + (wave + 8 * j) * 1024 + lane * 16;
            const float* tb = reinterpret_cast<const float*>(smem + TBL + par * 1024) + (bits & 3) * 16;
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(tb), t1 = *reinterpret_cast<const f32x4*>(tb + 4);
            const f32x4 t2 = *reinterpret_cast<const f32x4*>(tb + 8), t3 = *reinterpret_cast<const f32x4*>(tb + 12);
            float f[8];
            unpack8<T>(*reinterpret_cast<const U4*>(qp), f);
            f[0] = f[0] * t0[0] + t0[1]; f[1] = f[1] * t0[2] + t0[3];
            f[2] = f[2] * t1[0] + t1[1]; f[3] = f[3] * t1[2] + t1[3];
            f[4] = f[4] * t2[0] + t2[1]; f[5] = f[5] * t2[2] + t2[3];
            f[6] = f[6] * t3[0] + t3[1]; f[7] = f[7] * t3[2] + t3[3];
            if (p.a_gn_silu) {       // eight values in lockstep (common.h, gelu_erf_lockstep)
                float e[8];
                pin8(f);
#pragma unroll
                for (int i = 0; i < 8; ++i) e[i] = f[i] * -1.4426950408889634f;
                pin8(e);
#pragma unroll
                for (int i = 0; i < 8; ++i) e[i] = __builtin_amdgcn_exp2f(e[i]);
                pin8(e);
#pragma unroll
                for (int i = 0; i < 8; ++i) e[i] = 1.0f + e[i];
                pin8(e);
#pragma unroll
                for (int i = 0; i < 8; ++i) e[i] = __builtin_amdgcn_rcpf(e[i]);
                pin8(e);
#pragma unroll
                for (int i = 0; i < 8; ++i) f[i] = f[i] * e[i];
            }
            *reinterpret_cast<U4*>(qp) = pack8<T>(f);
        }
    };

    // ---- fragment read geometry.  Pixel block (j, h) of this wave: patch row 4 q + j (+ ky), pixels 16 h + l15 (+ kx); the key has
    // period 8, so the right half is the left half + 1 KiB
    int a_rd[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int px = l15 + kx;
        a_rd[kx] = (4 * q * PW + px) * 64 + ((lq ^ key8(px)) << 4);
    }
    const int b_rd = (wc * 64 + l15) * 64 + ((lq ^ key8(l15)) << 4);      // + nb KiB

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    U4 afr[4], bfr[4];

    // ---- prologue: patch of chunk 0 (+ its table)
    H5_STAMP(1);
    if (gnf) stage_t(0, 0);
#pragma unroll
    for (int j = 0; j < NPP; ++j) stage_p(0, j, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (gnf) {               // the first chunk's patch: every wave normalises its own pieces (its own table copy has landed too)
#pragma unroll
        for (int j = 0; j < NPP; ++j) gn_piece(j, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (g == 1) __builtin_amdgcn_s_barrier();           // waves 4-7 run half a phase behind their SIMD partners
    asm volatile("" ::: "memory");

    H5_STAMP(2);
    constexpr int PP0 = 2;                               // phases PP0 .. PP0 + NPP - 1 of a chunk stage a piece of the next patch
    // (the last chunk is a peeled copy of the loop body: no next patch to normalise, so its registers can hold the early residual)
    auto chunk_body = [&](int c, auto LASTc) {
        constexpr bool LAST = decltype(LASTc)::value;
        const int par = c & 1;
        const char* pa = smem + par * PATCHB;
        auto phase = [&](auto TAPc, auto SUBc) {
            constexpr int TAP = decltype(TAPc)::value, SUB = decltype(SUBc)::value, KY = TAP / 3, KX = TAP % 3;
            constexpr int TAP2 = (TAP + 2) % 9, PH = 2 * TAP + SUB, BUF = TAP % 3, BUF2 = TAP2 % 3;
            const int c2 = TAP + 2 >= 9 ? c + 1 : c;
            if constexpr (SUB == 0) {
                const char* pb = smem + W_BASE + BUF * WTAP + b_rd;
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) bfr[nb] = *reinterpret_cast<const U4*>(pb + nb * 1024);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) afr[j] = *reinterpret_cast<const U4*>(pa + a_rd[KX] + (j + KY) * PROWB + SUB * 1024);
            if constexpr (SUB == 0) stage_w(c2, TAP2, BUF2);
            if constexpr (PH >= PP0 && PH < PP0 + NPP) stage_p(c + 1, PH - PP0, par ^ 1);
            if constexpr (PH == 0) {
                if (gnf) stage_t(c + 1, par ^ 1);
            }
            if constexpr (SUB == 1) {
                // in flight by design: what this and the previous phase issued (one weight piece, up to two patch pieces, the table)
                constexpr int INFLIGHT = 1 + (PH >= PP0 && PH < PP0 + NPP ? 1 : 0) + (PH - 1 >= PP0 && PH - 1 < PP0 + NPP ? 1 : 0);
                if (PH == 1 && gnf) wait_vm<INFLIGHT + 1>();
                else wait_vm<INFLIGHT>();
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) acc[SUB * 4 + j][nb] = T::mfma16(bfr[nb], afr[j], acc[SUB * 4 + j][nb]);    // D[channel][pixel]
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (PH >= 6 && PH < 6 + NPP) {
                // the piece this wave staged in phase PH - 4 landed before the wait of phase PH - 1 at the latest
                if constexpr (!LAST) {
                    if (gnf) gn_piece(PH - 6, par ^ 1);
                }
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
        auto tap_body = [&](auto TAPc) { phase(TAPc, IC<0>{}); phase(TAPc, IC<1>{}); };
        tap_body(IC<0>{}); tap_body(IC<1>{}); tap_body(IC<2>{}); tap_body(IC<3>{}); tap_body(IC<4>{});
        tap_body(IC<5>{}); tap_body(IC<6>{}); tap_body(IC<7>{}); tap_body(IC<8>{});
#ifdef EDTR_STAMPS
        if (c == 0) H5_STAMP(6);
#endif
    };
    for (int c = 0; c < nchunk - 1; ++c) chunk_body(c, std::false_type{});
    chunk_body(nchunk - 1, std::true_type{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (g == 0) __builtin_amdgcn_s_barrier();           // re-align the two wave groups
    H5_STAMP(3);

    direct_epilogue<T, 8, RED>(p, acc, smem, smem_base, lane, wave, wc, (int64_t)m0 + 4 * q * p.OW, n0, img, tm);
    H5_STAMP(5); H5_STAMP(15);
}

// ------------------------------------------------------------------------------------------------------------------------------
// Tile 21 (round 6): tile 17 as a PERSISTENT kernel whose store phase runs under the next unit's multiply loop.
// A workgroup walks its units (its XCD's range, strided by the workgroups of that XCD) as ONE sequence of chunks: the last chunk
// of unit i stages the first patch and the first two weight slices of unit i + 1 exactly as a chunk stages its successor's, so
// there is no set-up / first-load phase between units.  After the last tap the 128 accumulators become the unit's FINISHED
// 16-bit output in 64 registers (bias added, GroupNorm sums taken, lanes P / P + 8 swapped so that a store instruction moves
// eight whole 128-byte lines: the arithmetic and the rounding of tile 17's finish16), the accumulators restart from zero (the
// first tap of the next unit multiplies into C = 0), and the 16 store instructions of the wave are issued ONE PER PHASE of the
// next unit's first chunk, next to that phase's DMA requests (loads, stores and LDS-DMA retire in issue order: the counted
// waits of those phases leave the stores in flight too).  The statistics of unit i meet in LDS behind the first phase of unit
// i + 1.  16-bit output without a residual or a time-embedding row only (bias rows live in LDS); the last unit of a workgroup
// flushes its stores at once.
// ------------------------------------------------------------------------------------------------------------------------------
namespace h1p {
using namespace h1;
constexpr int BIASL = RED + 4096;               // the launch's bias row (fp32, N <= 512)
constexpr int LDS_BYTES_P = BIASL + 2048;
}

template <int N> __device__ __forceinline__ void wait_vm_n() {
    static_assert(N >= 1 && N <= 12, "wait_vm_n");
    if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if constexpr (N == 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
}

template <typename T>
__global__ void __launch_bounds__(512, 1) igemm_halo512p_kernel(const edtr_igemm_params p) {
    using namespace h1p;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    H5_STAMP(0); H5_STAMP(14);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave >> 2, q = wave >> 1, wc = wave & 1;
    const int l15 = lane & 15, lq = lane >> 4;

    const int tw = p.OW >> 5, tpi = tw * (p.OH >> 4);
    const int nbn = p.N >> 7, nblk = (p.M / (p.OH * p.OW)) * tpi * nbn;
    // the units of this workgroup: XCD x (blockIdx mod 8) owns a contiguous range of units, its workgroups walk it with stride per
    const int per = (int)gridDim.x >> 3, xcd = (int)blockIdx.x & 7, jx = (int)blockIdx.x >> 3;
    const int qq = nblk >> 3, rr = nblk & 7;
    const int xs = xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq, xc = qq + (xcd < rr ? 1 : 0);
    if (jx >= xc) return;
    const int nmine = (xc - jx + per - 1) / per;

    const uint16_t* a1 = static_cast<const uint16_t*>(p.a1);
    const uint32_t smem_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    const u32x4 srd_a = srd_of(a1);
    const u32x4 srd_w = srd_of(p.w);
    const int Cin = p.C1, nchunk = Cin / CK;
    const bool gnf = p.a_gn != nullptr;
    const u32x4 srd_t = srd_of(gnf ? p.a_gn : reinterpret_cast<const float*>(a1));
    const bool gn_acc = p.gn_partial != nullptr;

    // unit geometry (wave-uniform): cur = the unit whose taps run, nxt = the unit whose first operands the last chunk stages
    struct Unit { int tm, n0, img, m0, sy0, sx0; };
    auto decode = [&](int k) {
        const int bid = xs + jx + k * per;
        Unit u;
        u.tm = bid / nbn;
        const int tn = bid - u.tm * nbn;
        u.img = u.tm / tpi;
        const int tr = u.tm - u.img * tpi, ty = tr / tw, tx = tr - ty * tw;
        u.n0 = tn * 128;
        u.sy0 = ty * 16 - 1;
        u.sx0 = tx * 32 - 1;
        u.m0 = (u.img * p.OH + ty * 16) * p.OW + tx * 32;
        return u;
    };

    // weight piece = wave (tile 17's permuted rows); the unit's n0 rides in the scalar offset
    uint32_t voff_w;
    {
        const int R = wave * 16 + (lane >> 2), slot = lane & 3;
        const int r = R & 15, nb = (R >> 4) & 3;
        const int n = (R & 64) + 32 * (nb >> 1) + 8 * (r >> 2) + 4 * (nb & 1) + (r & 3);
        const int c = slot ^ key8(R);
        voff_w = (uint32_t)(((int64_t)n * p.ldw + c * 8) * 2);
    }
    auto stage_w = [&](int n0, int chunk, int tap, int buf, bool valid) {
        dma(valid ? voff_w : kOob, srd_w, (uint32_t)(((int64_t)n0 * p.ldw + tap * Cin + chunk * CK) * 2), smem_base + W_BASE + buf * WTAP + wave * 1024);
    };
    uint32_t voff_p[NPP];
    uint32_t gn_bits = 0;
    auto set_patch = [&](const Unit& u, bool valid) {
        int ln = lane;
        asm volatile("" : "+v"(ln));           // (an opaque lane index: the unit-independent parts are recomputed per unit, not carried through the loop)
        gn_bits = 0;
#pragma unroll
        for (int j = 0; j < NPP; ++j) {
            const int uu = (wave + 8 * j) * 64 + ln, pp = uu >> 2, slot = uu & 3;
            const int py = pp / PW, px = pp - py * PW;
            const int iy = u.sy0 + py, ix = u.sx0 + px;
            const bool ok = valid && pp < PPIX && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
            const int c = slot ^ key8(px);
            voff_p[j] = ok ? (uint32_t)(((((int64_t)u.img * p.IH + iy) * p.IW + ix) * p.ld1 + c * 8) * 2) : kOob;
            if (ok) gn_bits |= (uint32_t)(8 | c) << (4 * j);
        }
    };
    auto stage_p = [&](int chunk, int j, int par) {
        dma(voff_p[j], srd_a, (uint32_t)(chunk * CK * 2), smem_base + par * PATCHB + (wave + 8 * j) * 1024);
    };
    auto stage_t = [&](int img, int chunk, int par, bool valid) {
        const uint32_t vo = (valid && lane < 16) ? (uint32_t)(lane * 16) : kOob;
        dma(vo, srd_t, (uint32_t)((img * Cin * 2 + chunk * CK * 2) * 4), smem_base + TBL + par * 1024);
    };
    auto gn_piece = [&](int j, int par) {
        const uint32_t bits = gn_bits >> (4 * j);
        if (bits & 8) {
            char* qp = smem + par * PATCHB + (wave + 8 * j) * 1024 + lane * 16;
            const float* tb = reinterpret_cast<const float*>(smem + TBL + par * 1024) + (bits & 3) * 16;
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(tb), t1 = *reinterpret_cast<const f32x4*>(tb + 4);
            const f32x4 t2 = *reinterpret_cast<const f32x4*>(tb + 8), t3 = *reinterpret_cast<const f32x4*>(tb + 12);
            float f[8];
            unpack8<T>(*reinterpret_cast<const U4*>(qp), f);
            f[0] = f[0] * t0[0] + t0[1]; f[1] = f[1] * t0[2] + t0[3];
            f[2] = f[2] * t1[0] + t1[1]; f[3] = f[3] * t1[2] + t1[3];
            f[4] = f[4] * t2[0] + t2[1]; f[5] = f[5] * t2[2] + t2[3];
            f[6] = f[6] * t3[0] + t3[1]; f[7] = f[7] * t3[2] + t3[3];
            if (p.a_gn_silu) {
                float e[8];
                pin8(f);
#pragma unroll
                for (int i = 0; i < 8; ++i) e[i] = f[i] * -1.4426950408889634f;
                pin8(e);
#pragma unroll
                for (int i = 0; i < 8; ++i) e[i] = __builtin_amdgcn_exp2f(e[i]);
                pin8(e);
#pragma unroll
                for (int i = 0; i < 8; ++i) e[i] = 1.0f + e[i];
                pin8(e);
#pragma unroll
                for (int i = 0; i < 8; ++i) e[i] = __builtin_amdgcn_rcpf(e[i]);
                pin8(e);
#pragma unroll
                for (int i = 0; i < 8; ++i) f[i] = f[i] * e[i];
            }
            *reinterpret_cast<U4*>(qp) = pack8<T>(f);
        }
    };

    int a_rd[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int px = l15 + kx;
        a_rd[kx] = (4 * q * PW + px) * 64 + ((lq ^ key8(px)) << 4);
    }
    const int b_rd = (wc * 64 + l15) * 64 + ((lq ^ key8(l15)) << 4);

    f32x4 acc[8][4];                               // (the first tap of every unit multiplies into C = 0)
    U4 afr[4], bfr[4];
    U4 pk[8][2];                                   // the previous unit's finished output, as its store instructions take it
#pragma unroll
    for (int i = 0; i < 8; ++i) { pk[i][0] = zero16(); pk[i][1] = zero16(); }

    Unit cur = decode(0), nxt = cur, prv = cur;
    bool has_next = false, have_prev = false;

    // ---- prologue (first unit only): bias row -> LDS, weight slices of taps 0 / 1, patch of chunk 0 (+ its table)
    stage_w(cur.n0, 0, 0, 0, true);
    stage_w(cur.n0, 0, 1, 1, true);
    set_patch(cur, true);
    H5_STAMP(1);
    if (gnf) stage_t(cur.img, 0, 0, true);
#pragma unroll
    for (int j = 0; j < NPP; ++j) stage_p(0, j, 0);
    if (tid < (p.N >> 2)) {
        f32x4 b = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (p.bias_n) b = *reinterpret_cast<const f32x4*>(p.bias_n + 4 * tid);
        *reinterpret_cast<f32x4*>(smem + BIASL + 16 * tid) = b;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (gnf) {
#pragma unroll
        for (int j = 0; j < NPP; ++j) gn_piece(j, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (g == 1) __builtin_amdgcn_s_barrier();           // waves 4-7 run half a phase behind their SIMD partners
    asm volatile("" ::: "memory");
    H5_STAMP(2);

    // The stores are buffer-addressed: lane offset + a wave-uniform scalar offset, and an out-of-range lane offset DROPS the lane — the first
    // unit of a workgroup has no predecessor, but its first chunk must issue the same 16 instructions (the counted waits count them).
    const u32x4 srd_o = srd_of(p.out);
    uint32_t so1 = kOob, so2 = kOob;
    auto store_piece = [&](const Unit& u, int k) {            // store instruction k of the wave: pixel block k >> 1, lanes' pixels pa (+ 8)
        const int mb = k >> 1;
        const int64_t mu = (int64_t)u.m0 + (4 * q + (mb & 3)) * p.OW + 16 * (mb >> 2);
        const uint32_t soff = (uint32_t)((mu * p.ldc + u.n0 + wc * 64) * 2);
        const u32x4 v = u32x4{pk[mb][k & 1].x, pk[mb][k & 1].y, pk[mb][k & 1].z, pk[mb][k & 1].w};
#ifdef H5P_NOSTORE     // (diagnostic builds only: what do the stores cost the loop?)
        asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" : : "v"(v), "v"(kOob), "s"(srd_o), "s"(soff) : "memory");
#else
        asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" : : "v"(v), "v"((k & 1) ? so2 : so1), "s"(srd_o), "s"(soff) : "memory");
#endif
    };
    float* const red = reinterpret_cast<float*>(smem + RED);
    auto ror8 = [&](const U4& v) {
        U4 r;
        r.x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.x, 0x128, 0xF, 0xF, false);
        r.y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.y, 0x128, 0xF, 0xF, false);
        r.z = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.z, 0x128, 0xF, 0xF, false);
        r.w = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.w, 0x128, 0xF, 0xF, false);
        return r;
    };
    auto sel = [&](bool c, const U4& a, const U4& b) {
        U4 r;
        r.x = c ? a.x : b.x; r.y = c ? a.y : b.y; r.z = c ? a.z : b.z; r.w = c ? a.w : b.w;
        return r;
    };
    // accumulators -> pk (bias, GroupNorm sums, one rounding); the wave's partial sums go to its own rows of `red`
    auto pack_unit = [&](const Unit& u) {
        // epilogue lane geometry (tile 17, finish16): instruction 1 writes pixel pa of a block, instruction 2 pixel pa + 8 — from an
        // opaque copy of the lane index, so that nothing of it is carried through the multiply loop
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int l15e = lane_e & 15, lqe = lane_e >> 4;
        const bool lo8 = l15e < 8;
        so1 = (uint32_t)((l15e & 7) * p.ldc + (lo8 ? 0 : 32) + 8 * lqe) * 2u;
        so2 = so1 + 16u * (uint32_t)p.ldc;
        const float alpha = p.alpha;
        const int nu = u.n0 + wc * 64;
        float cb[2][8], gs[2][8], gq[2][8];
#pragma unroll
        for (int hp = 0; hp < 2; ++hp) {
            const float* bl = reinterpret_cast<const float*>(smem + BIASL) + nu + 32 * hp + 8 * lqe;
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(bl), b1 = *reinterpret_cast<const f32x4*>(bl + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { cb[hp][j] = b0[j]; cb[hp][j + 4] = b1[j]; gs[hp][j] = 0.0f; gs[hp][j + 4] = 0.0f; gq[hp][j] = 0.0f; gq[hp][j + 4] = 0.0f; }
        }
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) {
            float f[2][8];
#pragma unroll
            for (int hp = 0; hp < 2; ++hp)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f[hp][i] = __builtin_fmaf(acc[mb][2 * hp][i], alpha, cb[hp][i]);
                    f[hp][i + 4] = __builtin_fmaf(acc[mb][2 * hp + 1][i], alpha, cb[hp][i + 4]);
                }
#pragma unroll
            for (int hp = 0; hp < 2; ++hp) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { gs[hp][i] += f[hp][i]; gq[hp][i] += f[hp][i] * f[hp][i]; }
                pin8(gs[hp]);
                pin8(gq[hp]);
            }
            const U4 h0 = pack8<T>(f[0]), h1 = pack8<T>(f[1]);
            const U4 got = ror8(sel(lo8, h1, h0));
            pk[mb][0] = sel(lo8, h0, got);
            pk[mb][1] = sel(lo8, got, h1);
        }
        if (gn_acc) {
#pragma unroll
            for (int hp = 0; hp < 2; ++hp) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { gs[hp][j] = row16_sum(gs[hp][j]); gq[hp][j] = row16_sum(gq[hp][j]); }
                if (l15e == 0) {
                    float* dst = red + (wave * 64 + 32 * hp + 8 * lqe) * 2;
#pragma unroll
                    for (int j = 0; j < 8; ++j) { dst[2 * j] = gs[hp][j]; dst[2 * j + 1] = gq[hp][j]; }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    };
    auto gn_publish = [&](const Unit& u) {                    // per-channel sums of the unit's 512 pixels -> its four 128-row slots (the first takes them)
        if (tid < 128) {
            const int wcc = tid >> 6, cl = tid & 63;
            float a = 0.0f, s = 0.0f;
#pragma unroll
            for (int k = 0; k < 4; ++k) { a += red[((2 * k + wcc) * 64 + cl) * 2]; s += red[((2 * k + wcc) * 64 + cl) * 2 + 1]; }
            float* dst = p.gn_partial + ((int64_t)(4 * u.tm) * gn_ld_of(p) + u.n0 + tid) * 2;
            dst[0] = a;
            dst[1] = s;
#pragma unroll
            for (int k = 1; k < 4; ++k) { dst[2 * k * (int64_t)gn_ld_of(p)] = 0.0f; dst[2 * k * (int64_t)gn_ld_of(p) + 1] = 0.0f; }
        }
    };

    constexpr int PP0 = 2;                               // phases PP0 .. PP0 + NPP - 1 of a chunk stage a piece of the next patch
    // the previous unit's 16 store instructions: the first STB at once behind the pack (their registers never enter the loop, whose first
    // phases hold the most: old output + new accumulators + fragments), the others one per phase ST0 .. of the unit's first chunk
#ifndef H5P_STB
#define H5P_STB 4
#endif
#ifndef H5P_ST0
#define H5P_ST0 1
#endif
    constexpr int STB = H5P_STB, ST0 = H5P_ST0, STN = 16 - STB;
    static_assert(ST0 + STN <= 18, "the stores of a unit leave within the next unit's first chunk");
    // PAR = LDS parity of the chunk's patch; LAST = last chunk of its unit (the "next chunk" is the next unit's first);
    // ST = first chunk of a unit that follows another: the first tap multiplies into C = 0 and the previous unit's stores are issued
    auto chunk_body = [&](int c, auto PARc, auto LASTc, auto STc) {
        constexpr int par = decltype(PARc)::value;
        constexpr bool LAST = decltype(LASTc)::value, ST = decltype(STc)::value;
        const char* pa = smem + par * PATCHB;
        auto phase = [&](auto TAPc, auto SUBc) {
            constexpr int TAP = decltype(TAPc)::value, SUB = decltype(SUBc)::value, KY = TAP / 3, KX = TAP % 3;
            constexpr int TAP2 = (TAP + 2) % 9, PH = 2 * TAP + SUB, BUF = TAP % 3, BUF2 = TAP2 % 3;
            if constexpr (SUB == 0) {
                const char* pb = smem + W_BASE + BUF * WTAP + b_rd;
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) bfr[nb] = *reinterpret_cast<const U4*>(pb + nb * 1024);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) afr[j] = *reinterpret_cast<const U4*>(pa + a_rd[KX] + (j + KY) * PROWB + SUB * 1024);
            if constexpr (SUB == 0) {
                if constexpr (TAP + 2 >= 9) {
                    if constexpr (LAST) stage_w(nxt.n0, 0, TAP2, BUF2, has_next);
                    else stage_w(cur.n0, c + 1, TAP2, BUF2, true);
                } else {
                    stage_w(cur.n0, c, TAP2, BUF2, true);
                }
            }
            if constexpr (PH >= PP0 && PH < PP0 + NPP) stage_p(LAST ? 0 : c + 1, PH - PP0, par ^ 1);      // (LAST: voff_p is the next unit's already)
            if constexpr (PH == 0) {
                if (gnf) {
                    if constexpr (LAST) stage_t(nxt.img, 0, par ^ 1, has_next);
                    else stage_t(cur.img, c + 1, par ^ 1, true);
                }
            }
            constexpr bool STH = ST && PH >= ST0 && PH < ST0 + STN, STP = ST && PH - 1 >= ST0 && PH - 1 < ST0 + STN;
            if constexpr (STH) store_piece(prv, STB + PH - ST0);
            if constexpr (SUB == 1) {
                // in flight by design: what this and the previous phase issued (one weight piece, up to two patch pieces, the table, up to two stores)
                // (+ in phase 1 the STB stores issued between the units: younger than everything phase 2 reads)
                constexpr int INFLIGHT = 1 + (PH >= PP0 && PH < PP0 + NPP ? 1 : 0) + (PH - 1 >= PP0 && PH - 1 < PP0 + NPP ? 1 : 0) + (STH ? 1 : 0) + (STP ? 1 : 0) +
                                         (ST && PH == 1 ? STB : 0);
                if (PH == 1 && gnf) wait_vm_n<INFLIGHT + 1>();
                else wait_vm_n<INFLIGHT>();
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    if constexpr (ST && TAP == 0) acc[SUB * 4 + j][nb] = T::mfma16(bfr[nb], afr[j], f32x4{0.0f, 0.0f, 0.0f, 0.0f});
                    else acc[SUB * 4 + j][nb] = T::mfma16(bfr[nb], afr[j], acc[SUB * 4 + j][nb]);
                }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (PH >= 6 && PH < 6 + NPP) {
                if (gnf) gn_piece(PH - 6, par ^ 1);         // (no next unit: gn_bits == 0)
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if constexpr (ST && PH == 0) {
                // every wave's partial sums of the previous unit are in `red` (each waited for its LDS writes before it came here)
                if (gn_acc && have_prev) gn_publish(prv);
            }
        };
        auto tap_body = [&](auto TAPc) { phase(TAPc, IC<0>{}); phase(TAPc, IC<1>{}); };
        tap_body(IC<0>{}); tap_body(IC<1>{}); tap_body(IC<2>{}); tap_body(IC<3>{}); tap_body(IC<4>{});
        tap_body(IC<5>{}); tap_body(IC<6>{}); tap_body(IC<7>{}); tap_body(IC<8>{});
    };

    using std::true_type;
    using std::false_type;
    for (int k = 0; k < nmine; ++k) {
        chunk_body(0, IC<0>{}, false_type{}, true_type{});
        for (int c = 1; c + 1 < nchunk; c += 2) {
            chunk_body(c, IC<1>{}, false_type{}, false_type{});
            chunk_body(c + 1, IC<0>{}, false_type{}, false_type{});
        }
        has_next = k + 1 < nmine;
        if (has_next) nxt = decode(k + 1);
        set_patch(nxt, has_next);                        // (the current unit's staging offsets are dead: its last patch is in LDS)
        chunk_body(nchunk - 1, IC<1>{}, true_type{}, false_type{});
#ifdef EDTR_STAMPS
        if (k < 6) H5_STAMP(8 + k);
#endif
        pack_unit(cur);
        prv = cur;
        cur = nxt;
        have_prev = true;
        if (has_next) {
#pragma unroll
            for (int i = 0; i < STB; ++i) store_piece(prv, i);
        }
    }
    H5_STAMP(3);
    // the last unit: its stores at once, its statistics behind a workgroup barrier
#pragma unroll
    for (int i = 0; i < 16; ++i) store_piece(prv, i);
    H5_STAMP(4);
    if (g == 0) __builtin_amdgcn_s_barrier();           // re-align the two wave groups
    if (gn_acc) {
        __syncthreads();
        gn_publish(prv);
    }
    H5_STAMP(5); H5_STAMP(15);
}

// ------------------------------------------------------------------------------------------------------------------------------
// Tile 20 (round 5): the halo tile for N % 160 == 0 — the UNet / ControlNet ResBlock convolutions of the 64 x 64 latent level
// (N = 320, M = 32768 at batch 8; reference model/unet.py:152,178).  The 128-column halo tiles pad N = 320 to three column tiles and
// 1.5 rounds of units, which is why these convolutions stayed on the 128 x 160 implicit-GEMM tile (tile 8: every input element staged
// nine times, 847 TFLOP/s).  Here the unit is 16 x 16 pixels x 160 channels: 128 x 2 = 256 units at batch 8, exactly one per CU.
// Wave w owns patch rows 2 w, 2 w + 1 (two 16-pixel blocks) x ALL 160 channels (ten 16-channel blocks: 80 accumulator registers);
// per 32-channel chunk and tap one phase of 20 MFMAs, 10 weight + 2 pixel fragment reads; waves w / w + 4 share a SIMD and run half
// a phase apart (the ping-pong of tiles 16 / 17).  Weight slices are 10 KiB (160 rows x 64 B; every wave issues two pieces per tap,
// pieces 10..15 are out-of-range and land in a dump KiB), ring of FOUR: the slice of tap t + 3 is requested behind the first barrier
// of phase t (every wave is then past its reads of tap t - 1, whose slot it takes), the wait in front of phase t's first barrier
// leaves only phase t - 1's requests in flight, i.e. retires tap t + 1's slice two phases after its request.
// LDS row R = 16 nb + r of a slice holds channel 40 (r >> 2) + 4 nb + (r & 3): a lane ends with FORTY consecutive channels of its pixel
// (five 16-byte runs), epilogue straight from the accumulators as in tile 17.
// ------------------------------------------------------------------------------------------------------------------------------
namespace h3 {
constexpr int PW = 18, PPIX = PW * PW, PROWB = PW * 64;
constexpr int NPP = 3;                          // patch pieces per wave and chunk: 8 x 3 = 24 one-KiB pieces, 21 real
constexpr int PATCHB = 21 * 1024;
constexpr int WSL = 10 * 1024;                  // one tap's [160][32] weight slice
constexpr int NRING = 4;
constexpr int W_BASE = 0, P_BASE = NRING * WSL, DUMP = P_BASE + 2 * PATCHB, RED = DUMP + 1024;
constexpr int LDS_BYTES = RED + 8 * 160 * 8;    // 94 KiB (GroupNorm scratch: [8 waves][160 channels][2] floats)
static_assert(PATCHB + 3 * PROWB < 65536 && NRING * WSL + 10 * 1024 < 65536, "ds_read immediates (the patch base sits in the lane register)");
}

template <typename T>
__global__ void __launch_bounds__(512, 1) igemm_halo160_kernel(const edtr_igemm_params p) {
    using namespace h3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave >> 2;
    const int l15 = lane & 15, lq = lane >> 4;

    const int tw = p.OW >> 4, tpi = tw * (p.OH >> 4);
    const int nbm = (p.M / (p.OH * p.OW)) * tpi, nbn = p.N / 160;
    int bid = blockIdx.x;
    {
        const int nblk = nbm * nbn, qq = nblk >> 3, r = nblk & 7, x = bid & 7, j = bid >> 3;
        bid = (x < r ? x * (qq + 1) : r * (qq + 1) + (x - r) * qq) + j;
    }
    const int tm = bid / nbn, tn = bid - tm * nbn;
    const int img = tm / tpi, tr = tm - img * tpi, ty = tr / tw, tx = tr - ty * tw;
    const int oy0 = ty * 16, ox0 = tx * 16, n0 = tn * 160;
    const int sy0 = oy0 - 1, sx0 = ox0 - 1;
    const int m0 = (img * p.OH + oy0) * p.OW + ox0;

    const uint16_t* a1 = static_cast<const uint16_t*>(p.a1);
    const uint32_t smem_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    const u32x4 srd_a = srd_of(a1);
    const u32x4 srd_w = srd_of(p.w);
    const int Cin = p.C1, nchunk = Cin / CK;

    // ten weight pieces per tap (LDS rows 16 piece ..): wave w requests piece w, waves 0 and 1 also pieces 8 and 9 — NOT two requests
    // from every wave with six of them empty: a DMA issue costs ~65 cycles whether or not it moves anything, and the part of a
    // phase that is not MFMAs (12 fragment reads, the requests, wait + barriers) must fit under the partner wave's 320 cycles of
    // MFMAs.  The counted waits therefore differ between waves 0 - 1 and 2 - 7 (a scalar branch per phase).
    const int nwp = wave < 2 ? 2 : 1;
    uint32_t voff_w[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int R = (wave + 8 * jj) * 16 + (lane >> 2), slot = lane & 3;
        const int r = R & 15, nb = R >> 4;
        const int n = n0 + 40 * (r >> 2) + 4 * nb + (r & 3);
        voff_w[jj] = R < 160 ? (uint32_t)(((int64_t)n * p.ldw + (slot ^ key8(R)) * 8) * 2) : kOob;
    }
    auto stage_w = [&](int chunk, int tap, int buf) {
        dma(chunk < nchunk ? voff_w[0] : kOob, srd_w, (uint32_t)((tap * Cin + chunk * CK) * 2), smem_base + W_BASE + buf * WSL + wave * 1024);
        if (nwp == 2) dma(chunk < nchunk ? voff_w[1] : kOob, srd_w, (uint32_t)((tap * Cin + chunk * CK) * 2), smem_base + W_BASE + buf * WSL + (wave + 8) * 1024);
    };
    stage_w(0, 0, 0);
    stage_w(0, 1, 1);
    stage_w(0, 2, 2);
    uint32_t voff_p[NPP];
#pragma unroll
    for (int j = 0; j < NPP; ++j) {
        const int u = (wave + 8 * j) * 64 + lane, pp = u >> 2, slot = u & 3;
        const int py = pp / PW, px = pp - py * PW;
        const int iy = sy0 + py, ix = sx0 + px;
        const bool ok = pp < PPIX && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
        voff_p[j] = ok ? (uint32_t)(((((int64_t)img * p.IH + iy) * p.IW + ix) * p.ld1 + (slot ^ key8(px)) * 8) * 2) : kOob;
    }
    auto stage_p = [&](int chunk, int j, int par) {
        const int piece = wave + 8 * j;
        dma(chunk < nchunk ? voff_p[j] : kOob, srd_a, (uint32_t)(chunk * CK * 2), smem_base + (piece < 21 ? P_BASE + par * PATCHB + piece * 1024 : DUMP));
    };

    // fragment reads: pixel block mb = patch row 2 wave + mb (+ ky), pixels l15 (+ kx); weight block nb = slice rows 16 nb + l15
    int a_rd[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int px = l15 + kx;
        a_rd[kx] = P_BASE + (2 * wave * PW + px) * 64 + ((lq ^ key8(px)) << 4);
    }
    const int b_rd = W_BASE + l15 * 64 + ((lq ^ key8(l15)) << 4);       // + slot WSL + nb KiB

    f32x4 acc[2][10];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 10; ++j) acc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

#pragma unroll
    for (int j = 0; j < NPP; ++j) stage_p(0, j, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (g == 1) __builtin_amdgcn_s_barrier();           // waves 4-7 run half a phase behind their SIMD partners
    asm volatile("" ::: "memory");

    // (chunk loop unrolled by four: patch parity and ring position (9 c) & 3 = c & 3 are compile-time)
    auto chunk_body = [&](int c, auto POSc) {
        constexpr int POS = decltype(POSc)::value, par = POS & 1, rb = POS;
        auto phase = [&](auto TAPc) {
            constexpr int TAP = decltype(TAPc)::value, KY = TAP / 3, KX = TAP % 3, TAP3 = (TAP + 3) % 9, PREV = (TAP + 8) % 9;
            const int c3 = TAP + 3 >= 9 ? c + 1 : c;
            U4 bfr[10], afr[2];
            const char* pb = smem + ((rb + TAP) & 3) * WSL + b_rd;
#pragma unroll
            for (int nb = 0; nb < 10; ++nb) bfr[nb] = *reinterpret_cast<const U4*>(pb + nb * 1024);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) afr[mb] = *reinterpret_cast<const U4*>(smem + par * PATCHB + a_rd[KX] + (mb + KY) * PROWB);
            // in flight by design: what the PREVIOUS phase requested behind its barrier (two weight pieces, a patch piece in phases 1..3)
            constexpr int INFL = 1 + (PREV >= 1 && PREV <= NPP ? 1 : 0);
            if (nwp == 2) wait_vm<INFL + 1>();
            else wait_vm<INFL>();
            __builtin_amdgcn_s_barrier();
            stage_w(c3, TAP3, (rb + TAP + 3) & 3);
            if constexpr (TAP >= 1 && TAP <= NPP) stage_p(c + 1, TAP - 1, par ^ 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 10; ++nb) acc[mb][nb] = T::mfma16(bfr[nb], afr[mb], acc[mb][nb]);    // D[channel][pixel]
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
        phase(IC<0>{}); phase(IC<1>{}); phase(IC<2>{}); phase(IC<3>{}); phase(IC<4>{}); phase(IC<5>{}); phase(IC<6>{}); phase(IC<7>{}); phase(IC<8>{});
    };
    for (int c0 = 0; c0 < nchunk; c0 += 4) {
        chunk_body(c0, IC<0>{});
        if (c0 + 1 < nchunk) chunk_body(c0 + 1, IC<1>{});
        if (c0 + 2 < nchunk) chunk_body(c0 + 2, IC<2>{});
        if (c0 + 3 < nchunk) chunk_body(c0 + 3, IC<3>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (g == 0) __builtin_amdgcn_s_barrier();           // re-align the two wave groups

    // ---- epilogue straight from the accumulators: acc[mb][nb][i] = pixel (row 2 wave + mb, column l15), channel n0 + 40 lq + 4 nb + i
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int l15e = lane_e & 15, lqe = lane_e >> 4;
    const float alpha = p.alpha;
    const bool gn_acc = p.gn_partial != nullptr;
    const int64_t mu0 = (int64_t)m0 + 2 * wave * p.OW;   // + mb OW: first pixel of block mb (wave-uniform)
    const int nl = n0 + 40 * lqe;                        // first of this lane's 40 channels
    const uint32_t lo_out = (uint32_t)(l15e * p.ldc + 40 * lqe), lo_res = (uint32_t)(l15e * p.ldr + 40 * lqe), lo_16 = (uint32_t)(l15e * p.ld16 + 40 * lqe);
    float* red = reinterpret_cast<float*>(smem + RED);
    auto finish = [&](auto out32_c, auto res_c, auto mir_c) {
        constexpr bool OUT32 = decltype(out32_c)::value, MIR = decltype(mir_c)::value && OUT32;
        constexpr int RES = decltype(res_c)::value;      // 0 none, 1 16-bit, 2 fp32
        U4 r16[2][5];
        if constexpr (RES == 1) {                        // all ten residual vectors of the lane at once: one exposure of the memory latency
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int k = 0; k < 5; ++k)
                    r16[mb][k] = ldg16(static_cast<const uint16_t*>(p.residual) + ((mu0 + mb * p.OW) * p.ldr + n0 + 8 * k) + lo_res);
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) {                    // channel run k: channels nl + 8 k .. + 7 = MFMA blocks 2 k, 2 k + 1
            float cb[8], gs[8], gq[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { cb[j] = 0.0f; gs[j] = 0.0f; gq[j] = 0.0f; }
            if (p.bias_n) {
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias_n + nl + 8 * k), b1 = *reinterpret_cast<const f32x4*>(p.bias_n + nl + 8 * k + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { cb[j] = b0[j]; cb[j + 4] = b1[j]; }
            }
            if (p.rowvec) {                              // the time-embedding row of the unit's image
                const float* rv = p.rowvec + (int64_t)img * p.rowvec_ld + nl + 8 * k;
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(rv), b1 = *reinterpret_cast<const f32x4*>(rv + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { cb[j] += b0[j]; cb[j + 4] += b1[j]; }
            }
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const int64_t mu = mu0 + mb * p.OW;
                float f[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f[i] = __builtin_fmaf(acc[mb][2 * k][i], alpha, cb[i]);
                    f[i + 4] = __builtin_fmaf(acc[mb][2 * k + 1][i], alpha, cb[i + 4]);
                }
                if constexpr (RES == 1) {
                    float rf[8];
                    unpack8<T>(r16[mb][k], rf);
#pragma unroll
                    for (int i = 0; i < 8; ++i) f[i] += rf[i];
                } else if constexpr (RES == 2) {
                    const float* rp = static_cast<const float*>(p.residual) + (mu * p.ldr + n0 + 8 * k) + lo_res;
                    const f32x4 q0 = *reinterpret_cast<const f32x4*>(rp), q1 = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll
                    for (int i = 0; i < 4; ++i) { f[i] += q0[i]; f[i + 4] += q1[i]; }
                }
                const int64_t oidx = mu * p.ldc + n0 + 8 * k;          // wave-uniform
                if constexpr (OUT32) {
                    float* o = static_cast<float*>(p.out) + oidx + lo_out;
                    *reinterpret_cast<f32x4*>(o) = f32x4{f[0], f[1], f[2], f[3]};
                    *reinterpret_cast<f32x4*>(o + 4) = f32x4{f[4], f[5], f[6], f[7]};
                    if constexpr (MIR) stg16(static_cast<uint16_t*>(p.out16) + (mu * p.ld16 + n0 + 8 * k) + lo_16, pack8<T>(f));
                } else {
                    stg16(static_cast<uint16_t*>(p.out) + oidx + lo_out, pack8<T>(f));
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) { gs[i] += f[i]; gq[i] += f[i] * f[i]; }
                pin8(gs);
                pin8(gq);
            }
            if (gn_acc) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { gs[j] = row16_sum(gs[j]); gq[j] = row16_sum(gq[j]); }
                if (l15e == 0) {
                    float* dst = red + (wave * 160 + 40 * lqe + 8 * k) * 2;
#pragma unroll
                    for (int j = 0; j < 8; ++j) { dst[2 * j] = gs[j]; dst[2 * j + 1] = gq[j]; }
                }
            }
        }
    };
    {
        using std::true_type;
        using std::false_type;
        const bool res32 = p.residual && p.residual_f32, res16 = p.residual && !p.residual_f32, mir = p.out16 != nullptr;
        if (!p.out_f32) {
            if (res16) finish(false_type{}, IC<1>{}, false_type{});
            else if (res32) finish(false_type{}, IC<2>{}, false_type{});
            else finish(false_type{}, IC<0>{}, false_type{});
        } else if (mir) {
            if (res32) finish(true_type{}, IC<2>{}, true_type{});
            else if (res16) finish(true_type{}, IC<1>{}, true_type{});
            else finish(true_type{}, IC<0>{}, true_type{});
        } else {
            if (res32) finish(true_type{}, IC<2>{}, false_type{});
            else if (res16) finish(true_type{}, IC<1>{}, false_type{});
            else finish(true_type{}, IC<0>{}, false_type{});
        }
    }
    if (gn_acc) {
        __syncthreads();
        if (tid < 160) {
            float a = 0.0f, sq = 0.0f;
#pragma unroll
            for (int w = 0; w < 8; ++w) { a += red[(w * 160 + tid) * 2]; sq += red[(w * 160 + tid) * 2 + 1]; }
            float* dst = p.gn_partial + ((int64_t)(2 * tm) * gn_ld_of(p) + n0 + tid) * 2;      // two 128-row slots per 256-pixel unit
            dst[0] = a;
            dst[1] = sq;
            dst[2 * (int64_t)gn_ld_of(p)] = 0.0f;
            dst[2 * (int64_t)gn_ld_of(p) + 1] = 0.0f;
        }
    }
}

template <typename T>
int launch_halo160(const edtr_igemm_params& p, hipStream_t stream) {
    static EdtrLdsOnce attr_set;
    if (int rc_ = edtr_lds_attr(reinterpret_cast<const void*>(&igemm_halo160_kernel<T>), h3::LDS_BYTES, attr_set)) return rc_;
    const int nbm = (p.M / (p.OH * p.OW)) * (p.OH >> 4) * (p.OW >> 4), nbn = p.N / 160;
    hipLaunchKernelGGL((igemm_halo160_kernel<T>), dim3(nbm * nbn), dim3(512), h3::LDS_BYTES, stream, p);
    EDTR_LAUNCH_CHECK();
    return EDTR_OK;
}

template <typename T>
int launch_halo512(const edtr_igemm_params& p, hipStream_t stream) {
    using namespace h1;
    static EdtrLdsOnce attr_set;
    if (int rc_ = edtr_lds_attr(reinterpret_cast<const void*>(&igemm_halo512_kernel<T>), LDS_BYTES, attr_set)) return rc_;
    const int nbm = (p.M / (p.OH * p.OW)) * (p.OH >> 4) * (p.OW >> 5), nbn = p.N >> 7;
    hipLaunchKernelGGL((igemm_halo512_kernel<T>), dim3(nbm * nbn), dim3(512), LDS_BYTES, stream, p);
    EDTR_LAUNCH_CHECK();
    return EDTR_OK;
}

template <typename T>
int launch_halo512p(const edtr_igemm_params& p, hipStream_t stream) {
    using namespace h1p;
    static EdtrLdsOnce attr_set;
    if (int rc_ = edtr_lds_attr(reinterpret_cast<const void*>(&igemm_halo512p_kernel<T>), LDS_BYTES_P, attr_set)) return rc_;
    const int64_t units = (int64_t)(p.M >> 9) * (p.N >> 7);
    int grid = edtr_cu_count() & ~7;                     // one workgroup per CU, a multiple of the eight XCDs
    if (grid > units) grid = (int)((units + 7) & ~7);
    hipLaunchKernelGGL((igemm_halo512p_kernel<T>), dim3(grid), dim3(512), LDS_BYTES_P, stream, p);
    EDTR_LAUNCH_CHECK();
    return EDTR_OK;
}

}  // namespace

// tile 21 = tile 17's shapes with a 16-bit output, no residual, no time-embedding row, an even number of 32-channel chunks
bool edtr_halo512p_ok(const edtr_igemm_params& p) {
    return edtr_halo512_ok(p) && !p.out_f32 && !p.out16 && !p.residual && !p.rowvec && p.N <= 512 && ((p.C1 / CK) & 1) == 0 &&
           (int64_t)p.M * p.ldc * 2 < 0xF0000000LL;      // (the output is written by buffer-addressed stores)
}

int edtr_launch_halo512p(const edtr_igemm_params& p, hipStream_t stream) {
    return p.dtype == EDTR_BF16 ? launch_halo512p<BF16>(p, stream) : launch_halo512p<F16>(p, stream);
}

// shape / option requirements of tile 17 (the caller has validated the generic edtr_igemm rules and buffer addressability)
bool edtr_halo512_ok(const edtr_igemm_params& p) {
    return p.OH > 0 && p.taps == 9 && p.stride == 1 && p.pad_t == 1 && p.pad_l == 1 && p.C2 == 0 && (p.C1 % CK) == 0 && !p.upsample2x &&
           p.OH == p.IH && p.OW == p.IW && (p.OH & 15) == 0 && (p.OW & 31) == 0 && p.Z == 1 && p.splitk <= 1 && p.act == EDTR_ACT_NONE &&
           !p.bias_m && (p.N & 127) == 0 && (p.n_valid == 0 || p.n_valid == p.N) && !p.vt_out && !p.row_stats && !p.ln_stats && !p.a_wrap &&
           p.M == (p.M / (p.OH * p.OW)) * p.OH * p.OW && (!p.rowvec || p.rows_per_image == p.OH * p.OW) &&
           (!p.residual || p.residual_f32 || (int64_t)p.M * p.ldr * 2 < 0xF0000000LL);      // (a 16-bit residual is fetched by buffer-addressed DMA)
}

int edtr_launch_halo512(const edtr_igemm_params& p, hipStream_t stream) {
    return p.dtype == EDTR_BF16 ? launch_halo512<BF16>(p, stream) : launch_halo512<F16>(p, stream);
}

// tile 20: 16 x 16-pixel units x 160 channels (no a_gn: the emitters fuse a GroupNorm input only into N <= 128 convolutions)
bool edtr_halo160_ok(const edtr_igemm_params& p) {
    return p.OH > 0 && p.taps == 9 && p.stride == 1 && p.pad_t == 1 && p.pad_l == 1 && p.C2 == 0 && (p.C1 % CK) == 0 && !p.upsample2x &&
           p.OH == p.IH && p.OW == p.IW && (p.OH & 15) == 0 && (p.OW & 15) == 0 && p.Z == 1 && p.splitk <= 1 && p.act == EDTR_ACT_NONE &&
           !p.bias_m && p.N % 160 == 0 && (p.n_valid == 0 || p.n_valid == p.N) && !p.vt_out && !p.row_stats && !p.ln_stats && !p.a_wrap && !p.a_gn &&
           p.M == (p.M / (p.OH * p.OW)) * p.OH * p.OW && (!p.rowvec || p.rows_per_image == p.OH * p.OW);
}

int edtr_launch_halo160(const edtr_igemm_params& p, hipStream_t stream) {
    return p.dtype == EDTR_BF16 ? launch_halo160<BF16>(p, stream) : launch_halo160<F16>(p, stream);
}

#ifdef EDTR_STAMPS
extern "C" int edtr_halo512_stamped(const edtr_igemm_params* p, void* stream) {      // stand-alone diagnostic build: no edtr_igemm in front
    if (p->tile == 20) return edtr_halo160_ok(*p) ? edtr_launch_halo160(*p, static_cast<hipStream_t>(stream)) : EDTR_E_UNSUPPORTED;
    if (p->tile == 21) return edtr_halo512p_ok(*p) ? edtr_launch_halo512p(*p, static_cast<hipStream_t>(stream)) : EDTR_E_UNSUPPORTED;
    if (!edtr_halo512_ok(*p)) return EDTR_E_UNSUPPORTED;
    return edtr_launch_halo512(*p, static_cast<hipStream_t>(stream));
}
#endif
