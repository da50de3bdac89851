// Fused ConvLSTM gate convolution on 128 x 128 WAVE tiles, persistent workgroups (round 6).  Included by conv_fwd.hip inside its
// anonymous namespace (uses ConvArgs of that file).  e2vid/model/submodules.py:175-214.
//
// Why: the 64 x 64 wave tile of conv3x3_halo_tile issues one ds_read_b128 per MFMA (round-5 PMC: 1.1 LDS instructions per MFMA,
// matrix pipe 46 % busy, waves 40 % in issue stalls).  Here ONE wave per SIMD owns a 128-pixel x 128-gate-column accumulator
// block (16 x 32x32 tiles = 256 AGPRs of the unified 512-entry file): 8 fragment reads per 16 MFMAs = 0.5 per MFMA, and the
// workgroup tile is 256 pixels x 256 gate columns (64 hidden channels), so a K-slab moves 32 KB of weights + a third of a 40 KB
// halo for 256 MFMAs (0.71 KB of L2 -> LDS traffic per MFMA; the 256 x 128 tile moves 0.92).
//
// Nothing hides behind a second wave on the SIMD, so (1) the instruction stream of the K loop is laid out by hand and (2) the
// work around the K loop is taken off the tile's critical path:
//  (1) every instruction of the K loop is a volatile asm statement (MFMAs with "+a" accumulators, ds_read_b128, s_waitcnt,
//      s_barrier) or an LDS-DMA builtin -- hipcc keeps their relative order and only allocates registers.  A K-slab (64 k) is
//      four groups G0..G3 of 16 MFMAs; gap m of a group (the issue slots behind MFMA m) carries
//          m = 0..7   one fragment read of the NEXT k-step into the other fragment buffer (W0 P0 P1 P2 P3 W1 W2 W3),
//          m = 8..15  at most one LDS-DMA piece (1 KB),
//      and the group starts with lgkmcnt(0), which the last read precedes by eight MFMAs (256 cycles).  The barrier that opens
//      slab s + 1 sits between G2 and G3 of slab s: behind it G3's gaps read k-step 0 of slab s + 1 (across the slab boundary) and
//      issue the weight slab s + 2 into the stage slab s has just finished reading; the halo of macro step j + 1 is issued in
//      G0..G2 of slab (j, dx = 0).  vmcnt is counted: 10 halo pieces stay in flight across the barrier of (j, 0).
//      In-kernel stamps of the first form (one tile per workgroup, profiles/r06_w128_v1_stamps.txt): clean groups run 524-532
//      cycles per 16 MFMAs (512 = the pipe).
//  (2) the same stamps: address set-up 8-12 k cycles, first operands 2-5 k, cell-update epilogue 33-36 k per tile against a K loop
//      of 43 / 90 / 174 k on the three E2VID levels -- 38 % of the launch outside the K loop.  Therefore:
//      * PERSISTENT workgroups (one per CU) walk a static, host-built tile list (longest-K tiles first, greedy onto the least
//        loaded CU of the tile's XCD); problem records, the tile list and the gate biases live in LDS, a tile's set-up is
//        ~200 VALU instructions with no memory round trip;
//      * the cell state is kept in the KERNEL'S OWN layout ("w128-tiled": [tile][wave][16][lane][4] fp32 = the accumulator
//        layout), so the previous cell arrives with 16 coalesced 16-byte loads per lane issued at the start of the tile's K loop
//        and the new cell leaves with 16 coalesced stores -- no LDS transposition, no scattered 4-byte loads;
//      * the accumulators start at zero for free (first k-step: srcC = 0) and the gate bias enters the cell update as the addend
//        of the FMA that scales the exponent argument (table pre-scaled in LDS);
//      * the hidden state leaves as 8-byte pieces after two v_permlane32_swap per (pixel block, gate block): no LDS image;
//      * the next tile's first halo + two weight slabs are issued BEFORE the cell update of the current tile and land under it.
// LDS: [halo 0][halo 1] 2 x 40 KB, [weights 0][weights 1] 2 x 32 KB = 144 KB operands + 8 KB tables, one workgroup per CU.
// Requires: 3 x 3, dil 1, stride 1, pad 1, Cin % 64 == 0, Cout % 256 == 0, H >= 8, 32-bit buffer offsets (host: w128_eligible).
// cache policy of the streamed state (gfx950 buffer aux: 2 = nt): previous-cell loads, new-cell stores, hidden stores (measurement knobs)
#ifndef W128_CELL_LD_AUX
#define W128_CELL_LD_AUX 0
#endif
#ifndef W128_CELL_ST_AUX
#define W128_CELL_ST_AUX 0
#endif
#ifndef W128_H_ST_AUX
#define W128_H_ST_AUX 0
#endif
#ifndef W128_EARLY_CELL
#define W128_EARLY_CELL 0   // 1: previous-cell loads at the start of the tile's K loop (64 registers live across it)
#endif
#ifndef W128_E1_BLOCK
#define W128_E1_BLOCK 1    // cell-update blocks of 4 cells scheduled together (1, 2 or 4): instruction-level parallelism against registers
#endif
#ifndef W128_ABL
#define W128_ABL 0     // debug (tools/bench_lstm_group.py): 8192 = s_memtime stamps printed by the host after the launch
#endif
constexpr int W128_HROWS = 320;
constexpr int W128_HALO_BYTES = W128_HROWS * 128;         // 40 960
constexpr int W128_WST_BYTES = 256 * 128;                 // 32 768
constexpr int W128_OPER = 2 * W128_HALO_BYTES + 2 * W128_WST_BYTES;   // 147 456
constexpr int W128_BIAS_FLOATS = 1792;                    // 4 C summed over the (at most three) problems: 256 + 512 + 1024
constexpr int W128_MAX_LIST = 124;                        // tiles per workgroup (B = 8 at 440 x 640: 15)
constexpr int W128_LDS = W128_OPER + W128_BIAS_FLOATS * 4 + 3 * 128 + (W128_MAX_LIST + 4) * 4;

// per-problem record in LDS (32 ints): 0-1 in, 2-3 packed weights, 4-5 previous cell, 6-7 cell, 8-9 hidden (pointers lo / hi),
// 10 input bytes, 11 hidden bytes, 12 cell bytes, 13 input pixel stride, 14 hidden pixel stride, 15 Kpad, 16 H, 17 W, 18 Cin, 19 C,
// 20 M, 21 tiles_n, 22 / 23 exact reciprocals of W and W + 1, 25 first float of the bias table, 26 has a previous cell
struct W128Group {
    ConvArgs a[3];
    const int* sched;        // [gridDim.x][sched_stride]: (problem << 24) | tile, -1 = end; debug stamps behind it
    int sched_stride, n;
};

constexpr bool W128_STAMP = (W128_ABL & 8192) != 0;
#define W128_STAMP_TAKE() do { if constexpr (W128_STAMP) asm volatile("s_memtime %0" : "=s"(tnow)); } while (0)
#define W128_STAMP_ADD(K) do { if constexpr (W128_STAMP) { tacc[K] += (unsigned)tnow - tlast; tlast = (unsigned)tnow; } } while (0)

template <typename F, int... Is>
__device__ __forceinline__ void w128_for(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
#define W128_FOR(N, VAR, ...) w128_for([&](auto VAR) __attribute__((always_inline)) __VA_ARGS__, std::make_integer_sequence<int, N>{})
template <int V> using w128_c = std::integral_constant<int, V>;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

__device__ __forceinline__ int w128_sread(const int* p) { return __builtin_amdgcn_readfirstlane(*p); }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t w128_rsrc(const int* pr, int lo, int bytes) {
    const uintptr_t base = ((uintptr_t)(unsigned)w128_sread(pr + lo + 1) << 32) | (uintptr_t)(unsigned)w128_sread(pr + lo);
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv3x3_lstm_w128_kernel(W128Group g) {
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
    constexpr int NWAVES = 4, MT = 4, NT = 4;
    constexpr int H_INSTR = W128_HROWS / 8 / NWAVES;     // 10 halo pieces per wave and macro step
    constexpr int B_INSTR = 256 * 8 / 64 / NWAVES;       // 8 weight pieces per wave and slab
    constexpr int HALO_BYTES = W128_HALO_BYTES, WST = W128_WST_BYTES;
    constexpr float L2E = 1.4426950408889634f;

    float* lbias = reinterpret_cast<float*>(smem + W128_OPER);                              // pre-scaled gate biases, row n' = 4 hc + gate
    int* lprob = reinterpret_cast<int*>(smem + W128_OPER + W128_BIAS_FLOATS * 4);           // 3 x W128Prob
    int* llist = lprob + 3 * 32;                                                            // this workgroup's tile list

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int p31 = lane & 31, hi = lane >> 5;

    unsigned long long tnow = 0; unsigned tlast = 0; unsigned tacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned tsetup = 0, tfirst = 0;
    W128_STAMP_TAKE(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tlast = (unsigned)tnow;

    // ---- once per workgroup: problem records, tile list and bias tables into LDS
    {
        int boff = 0;
        for (int p = 0; p < g.n; ++p) {
            const ConvArgs& a = g.a[p];
            if (tid == 0) {
                int* d = lprob + p * 32;
                d[0] = (int)(uintptr_t)a.in; d[1] = (int)((uintptr_t)a.in >> 32);
                d[2] = (int)(uintptr_t)a.w; d[3] = (int)((uintptr_t)a.w >> 32);
                d[4] = (int)(uintptr_t)a.lstm_prev; d[5] = (int)((uintptr_t)a.lstm_prev >> 32);
                d[6] = (int)(uintptr_t)a.lstm_cell; d[7] = (int)((uintptr_t)a.lstm_cell >> 32);
                d[8] = (int)(uintptr_t)a.lstm_h; d[9] = (int)((uintptr_t)a.lstm_h >> 32);
                d[10] = (int)((((long long)a.B * a.H * a.W - 1) * a.in_pix_stride + a.Cin) * 2);
                d[11] = (int)((((long long)a.M - 1) * a.lstm_h_stride + a.lstm_C) * 2);
                d[12] = a.tiles_m * a.tiles_n * 65536;
                d[13] = (int)a.in_pix_stride; d[14] = (int)a.lstm_h_stride; d[15] = a.Kpad;
                d[16] = a.H; d[17] = a.W; d[18] = a.Cin; d[19] = a.lstm_C; d[20] = a.M; d[21] = a.tiles_n;
                d[22] = (int)a.mg_w; d[23] = (int)a.mg_wd; d[24] = 0; d[25] = boff; d[26] = a.lstm_prev ? 1 : 0;
            }
            const int C = a.lstm_C;
            for (int n = tid; n < 4 * C; n += 256) {             // n = 4 hc + gate <- Conv2d order gate * C + hc, pre-scaled for exp2
                const int hc = n >> 2, gate = n & 3;
                const float b = a.bias ? a.bias[gate * C + hc] : 0.0f;
                lbias[boff + n] = gate == 3 ? 2.0f * L2E * b : -L2E * b;
            }
            boff += 4 * C;
        }
        for (int k = tid; k < W128_MAX_LIST + 1; k += 256)
            llist[k] = k < g.sched_stride ? g.sched[(size_t)blockIdx.x * g.sched_stride + k] : -1;
    }
    __syncthreads();

    // ---- constant per workgroup: weight fragment addresses (gate rows of this wave), k-step ks of a row reads chunk (2 ks + half) ^ sw
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    uint32_t wa[NT][4];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int r = wn * 128 + j * 32 + p31;
        const uint32_t sw = (uint32_t)((r >> 1) & 7);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            wa[j][ks] = lds0 + (uint32_t)(2 * HALO_BYTES + r * 128) + ((((uint32_t)(ks * 2) + (uint32_t)hi) ^ sw) << 4);
            asm volatile("" : "+v"(wa[j][ks]));
        }
    }
    const unsigned oob = 0x80000000u;

    // ---- state carried from tile to tile
    f32x16_t acc[16];                                    // tile t = i*4 + j: pixels i*32.. x gate rows j*32.. of the wave's block
    f32x4_t cellreg[16];                                  // previous cell of the current tile, then its new cell (r = i*4 + j, [q])
    bf16x8_t fp[2][MT], fw[2][NT];                       // fragment double buffer: pixels / weights

    // per-tile values (set by `setup`)
    int hy[H_INSTR], hoff[H_INSTR], boff[B_INSTR];
    uint32_t pa[MT][3];                                   // pixel fragment address of k-step 0 per dx (k-step ks: ^ (ks << 5))
    __amdgpu_buffer_rsrc_t rsA, rsB, rsH, rsC, rsP;
    int Cin_s = 0, H_s = 0, ips_s = 0, W_s = 0, nch = 0, NJ = 0, M_s = 0;
    int hvoff[MT];                                        // hidden-store byte offset of pixel block i (0x80000000 = row >= M)
    int cell_voff = 0;                                    // byte offset of this lane in the wave's 16 KB block of the tile's cell state
    uint32_t bias_addr = 0;                               // LDS address of this lane's first bias quadruple

    auto setup = [&](int entry) __attribute__((always_inline)) {
        // opaque copies: everything below is recomputed per tile.  Left visible, hipcc hoists the lane-only parts (~100 values) out
        // of the tile loop and keeps them live across the K loop -> spills.
        int lane_o = lane, wave_o = wave;
        asm volatile("" : "+v"(lane_o), "+s"(wave_o));
        const int lrow = lane_o >> 3, slot = lane_o & 7, p31 = lane_o & 31, hi = lane_o >> 5, wave = wave_o, wm = wave_o >> 1, wn = wave_o & 1;
        const int p = entry >> 24, bid = entry & 0xffffff;
        const int* pr = lprob + p * 32;
        const int tiles_n = w128_sread(pr + 21);
        const int tile_n = bid % tiles_n, tile_m = bid / tiles_n;
        const int m0 = tile_m * 256, n0 = tile_n * 256;
        H_s = w128_sread(pr + 16); W_s = w128_sread(pr + 17); Cin_s = w128_sread(pr + 18); M_s = w128_sread(pr + 20);
        ips_s = w128_sread(pr + 13);
        const int hstride = w128_sread(pr + 14), Kpad = w128_sread(pr + 15);
        const unsigned mg_w = (unsigned)w128_sread(pr + 22), mg_wd = (unsigned)w128_sread(pr + 23);
        nch = Cin_s >> 6; NJ = 3 * nch;
        const int W = W_s, wd = W + 1;
        rsA = w128_rsrc(pr, 0, w128_sread(pr + 10));
        rsB = w128_rsrc(pr, 2, 0x7ffffff0);
        rsH = w128_rsrc(pr, 8, w128_sread(pr + 11));
        rsC = w128_rsrc(pr, 6, w128_sread(pr + 12));
        rsP = w128_rsrc(pr, 4, w128_sread(pr + 12));
        const int hw = H_s * W;
        const int b0 = m0 / hw, rem0 = m0 - b0 * hw;
        const int oy0 = (int)__umulhi((unsigned)rem0, mg_w), ox0 = rem0 - oy0 * W;
        const int L0 = (W - ox0 < 256) ? W - ox0 : 256;
        // halo DMA geometry (as conv3x3_halo_tile): lane (lrow, slot) of piece q writes halo row q*8 + lrow, 16-byte slot `slot`
#pragma unroll
        for (int i = 0; i < H_INSTR; ++i) {
            const int h = (wave * H_INSTR + i) * 8 + lrow;
            const int hp = h - 1;
            // branch-free (a divergent if / else costs an exec save / restore per piece): first image row of the tile, or segment q behind it
            const int h2 = hp - (L0 + 1);
            const int q = (int)__umulhi((unsigned)(h2 < 0 ? 0 : h2), mg_wd), r = h2 - q * wd;
            const bool first = h2 < 0;
            const int m_seg = first ? m0 : m0 + L0 + q * W, px = first ? ox0 + hp : r, drow = first ? 0 : q + 1;
            const bool valid = m_seg < M_s && (m_seg == m0 || m_seg - m0 < 256) && (unsigned)px < (unsigned)W;
            int oy = oy0 + drow;
            const int grow = b0 * H_s + oy;               // row index over the whole batch
            if (oy >= H_s) oy -= H_s;                     // a tile spans < 8 image rows (host: H >= 8)
            hy[i] = valid ? oy : -0x4000;
            hoff[i] = valid ? ((grow * W + px) * ips_s * 2) + (slot ^ ((h >> 1) & 7)) * 16 : 0;
        }
#pragma unroll
        for (int i = 0; i < B_INSTR; ++i) {
            const int r = (wave * B_INSTR + i) * 8 + lrow;
            boff[i] = ((n0 + r) * Kpad + (slot ^ ((r >> 1) & 7)) * 8) * 2;
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int r = wm * 128 + i * 32 + p31;
            const int t = r - L0, q = (int)__umulhi((unsigned)(t < 0 ? 0 : t), mg_w), rr = t - q * W;
            const int hr = t < 0 ? r : L0 + 1 + q * wd + rr;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int h = hr + dx;
                pa[i][dx] = lds0 + (uint32_t)h * 128 + ((((uint32_t)hi) ^ (uint32_t)((h >> 1) & 7)) << 4);
            }
            const int m = m0 + r;
            hvoff[i] = m < M_s ? (m * hstride + (n0 >> 2) + wn * 32 + 4 * hi) * 2 : (int)oob;
        }
        cell_voff = (bid * 4 + wave) * 16384 + lane_o * 16;
        bias_addr = lds0 + (uint32_t)(W128_OPER + (w128_sread(pr + 25) + n0 + wn * 128 + 4 * hi) * 4);
        return w128_sread(pr + 26);                       // has_prev
    };

    // ---- LDS-DMA pieces (the VALU of a halo piece is volatile asm too: left to hipcc it is hoisted in front of the slab's first MFMA)
    auto halo_piece = [&](auto par_c, auto i_c, int ddy, int tapoff) __attribute__((always_inline)) {
        constexpr int par = decltype(par_c)::value, i = decltype(i_c)::value;
        unsigned voff;
        asm volatile("v_add_u32 %0, %1, %2\n\tv_cmp_gt_u32 vcc, %3, %0\n\tv_add_u32 %0, %4, %5\n\tv_cndmask_b32 %0, %6, %0, vcc"
                     : "=&v"(voff) : "v"(hy[i]), "s"(ddy), "s"(H_s), "v"(hoff[i]), "s"(tapoff), "v"(oob) : "vcc");
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(smem + par * HALO_BYTES + (wave * H_INSTR + i) * 1024),
                                                 16, voff, 0, 0, 0);
    };
    auto w_piece = [&](auto st_c, auto i_c, int koff) __attribute__((always_inline)) {
        constexpr int st = decltype(st_c)::value, i = decltype(i_c)::value;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(smem + 2 * HALO_BYTES + st * WST + (wave * B_INSTR + i) * 1024),
                                                 16, (unsigned)boff[i], koff, 0, 0);
    };
    // byte offset along K of slab (dy, cc, dx); slabs past the end re-fetch the last one (never read)
    int koff_last = 0;
    auto slab_koff = [&](int dy, int cc, int dx) {
        const int k = ((dy * 3 + dx) * Cin_s + cc * 64) * 2;
        return k < koff_last ? k : koff_last;
    };

    // the pixel fragment address of k-step KS is built by a v_xor in front of the read (12 address registers instead of 48)
#define W128_RD_P(BUF, I, DX, KS, OFF) { uint32_t t_; asm volatile("v_xor_b32 %1, %4, %2\n\tds_read_b128 %0, %1 offset:%3" : "=v"(fp[BUF][I]), "=&v"(t_) : "v"(pa[I][DX]), "n"(OFF), "n"((KS) << 5) : "memory"); }
#define W128_RD_P0(BUF, I, DX, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fp[BUF][I]) : "v"(pa[I][DX]), "n"(OFF) : "memory")
#define W128_RD_W(BUF, J, KS, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fw[BUF][J]) : "v"(wa[J][KS]), "n"(OFF) : "memory")
    // read piece q (0..7) of k-step KS of the slab (halo offset HOFF, tap DX, weight stage offset WOFF) into fragment buffer BUF
    auto frag_read = [&fp, &fw, &pa, &wa](auto buf_c, auto q_c, auto dx_c, auto ks_c, auto hoff_c, auto woff_c) __attribute__((always_inline)) {
        constexpr int BUF = decltype(buf_c)::value, q = decltype(q_c)::value, DX = decltype(dx_c)::value, KS = decltype(ks_c)::value;
        constexpr int HOFF = decltype(hoff_c)::value, WOFF = decltype(woff_c)::value;
        if constexpr (q == 0) W128_RD_W(BUF, 0, KS, WOFF);
        else if constexpr (q <= 4) {
            if constexpr (KS == 0) W128_RD_P0(BUF, q - 1, DX, HOFF);
            else W128_RD_P(BUF, q - 1, DX, KS, HOFF);
        }
        else W128_RD_W(BUF, q - 4, KS, WOFF);
    };
    // MFMA m of a group on fragment buffer BUF: m = j*4 + i (weights are the A operand: a lane holds gate rows of ONE pixel)
    auto mma = [&acc, &fp, &fw](auto buf_c, auto m_c) __attribute__((always_inline)) {
        constexpr int BUF = decltype(buf_c)::value, m = decltype(m_c)::value, j = m >> 2, i = m & 3;
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i * 4 + j]) : "v"(fw[BUF][j]), "v"(fp[BUF][i]));
    };
    auto mma_first = [&acc, &fp, &fw](auto buf_c, auto m_c) __attribute__((always_inline)) {   // first k-step of a tile: C = 0
        constexpr int BUF = decltype(buf_c)::value, m = decltype(m_c)::value, j = m >> 2, i = m & 3;
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(acc[i * 4 + j]) : "v"(fw[BUF][j]), "v"(fp[BUF][i]));
    };

    int dy_c = 0, cc_c = 0, dy_n = 0, cc_n = 0;          // (dy, chunk) of macro steps j and j + 1
    bool first_slab = false;
    // one K-slab: macro step j of parity PAR, tap DX
    auto slab = [&](auto par_c, auto dx_c) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_c)::value, DX = decltype(dx_c)::value;
        constexpr int HOFF = PAR * HALO_BYTES, WSTAGE = (PAR + DX) & 1, WOFF = WSTAGE * WST;
        constexpr int nPAR = (DX == 2) ? (PAR ^ 1) : PAR, nDX = (DX + 1) % 3;
        constexpr int nHOFF = nPAR * HALO_BYTES, nWOFF = (WSTAGE ^ 1) * WST;
        using cDX = w128_c<DX>; using cH = w128_c<HOFF>; using cW = w128_c<WOFF>;
        // halo of macro step j + 1 (buffer PAR ^ 1): all ten pieces in slab (j, 0), G0 / G1 / G2 = 4 / 3 / 3; past the tile's last macro
        // step every row is out of range (no traffic)
        const int ddy_n = dy_n < 3 ? dy_n - 1 : 0x2000;
        const int tap_n = ((dy_n - 1) * W_s * ips_s + cc_n * 64) * 2;
        // weight slab s + 2 -> the stage this slab reads (free behind the barrier): (j, DX + 2) or (j + 1, DX - 1)
        const int koff2 = (DX == 0) ? slab_koff(dy_c, cc_c, 2) : slab_koff(dy_n, cc_n, DX - 1);
        // G0: MFMAs on buffer 0, reads of k-step 1 into buffer 1
        auto g0_fill = [&](auto m) __attribute__((always_inline)) {
            if constexpr (m < 8) frag_read(w128_c<1>{}, m, cDX{}, w128_c<1>{}, cH{}, cW{});
            else if constexpr (DX == 0 && (m & 1) == 0) halo_piece(w128_c<PAR ^ 1>{}, w128_c<(m - 8) / 2>{}, ddy_n, tap_n);
        };
        if (PAR == 0 && DX == 0 && first_slab) { W128_FOR(16, m, { mma_first(w128_c<0>{}, m); g0_fill(m); }); }
        else { W128_FOR(16, m, { mma(w128_c<0>{}, m); g0_fill(m); }); }
        W128_STAMP_TAKE();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        W128_STAMP_ADD(DX * 4 + 0);
        // G1: buffer 1, reads of k-step 2 into buffer 0
        W128_FOR(16, m, {
            mma(w128_c<1>{}, m);
            if constexpr (m < 8) frag_read(w128_c<0>{}, m, cDX{}, w128_c<2>{}, cH{}, cW{});
            else if constexpr (DX == 0 && (m & 1) == 0 && m < 14) halo_piece(w128_c<PAR ^ 1>{}, w128_c<4 + (m - 8) / 2>{}, ddy_n, tap_n);
        });
        W128_STAMP_TAKE();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        W128_STAMP_ADD(DX * 4 + 1);
        // G2: buffer 0, reads of k-step 3 into buffer 1
        W128_FOR(16, m, {
            mma(w128_c<0>{}, m);
            if constexpr (m < 8) frag_read(w128_c<1>{}, m, cDX{}, w128_c<3>{}, cH{}, cW{});
            else if constexpr (DX == 0 && (m & 1) == 0 && m < 14) halo_piece(w128_c<PAR ^ 1>{}, w128_c<7 + (m - 8) / 2>{}, ddy_n, tap_n);
        });
        // slab s + 1 landed (this wave's pieces), every wave is done reading slab s
        W128_STAMP_TAKE();
        if constexpr (DX == 0) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        W128_STAMP_ADD(DX * 4 + 2);
        // G3: buffer 1, reads of k-step 0 of slab s + 1 into buffer 0, weight slab s + 2
        W128_FOR(16, m, {
            mma(w128_c<1>{}, m);
            if constexpr (m < 8) frag_read(w128_c<0>{}, m, w128_c<nDX>{}, w128_c<0>{}, w128_c<nHOFF>{}, w128_c<nWOFF>{});
            else w_piece(w128_c<WSTAGE>{}, w128_c<m - 8>{}, koff2);
        });
        W128_STAMP_TAKE();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        W128_STAMP_ADD(DX * 4 + 3);
        if (PAR == 0 && DX == 0) first_slab = false;
    };
    auto macro_step = [&](auto par_c) __attribute__((always_inline)) {
        dy_n = dy_c; cc_n = cc_c + 1;
        if (cc_n == nch) { cc_n = 0; ++dy_n; }
        slab(par_c, w128_c<0>{});
        slab(par_c, w128_c<1>{});
        slab(par_c, w128_c<2>{});
        dy_c = dy_n; cc_c = cc_n;
    };
    // first halo + weight slabs 0, 1 of the tile just set up (all operand buffers are free: behind a barrier every wave has passed)
    auto fill = [&]() __attribute__((always_inline)) {
        koff_last = ((8 * Cin_s) + (nch - 1) * 64) * 2;
        W128_FOR(H_INSTR, i, { halo_piece(w128_c<0>{}, i, -1, (-W_s * ips_s) * 2); });
        W128_FOR(B_INSTR, i, { w_piece(w128_c<0>{}, i, slab_koff(0, 0, 0)); });
        W128_FOR(B_INSTR, i, { w_piece(w128_c<1>{}, i, slab_koff(0, 0, 1)); });
    };

    // ---- tile loop
    int li = 0;
    int entry = __builtin_amdgcn_readfirstlane(llist[0]);
    int has_prev = 0;
    if (entry >= 0) { has_prev = setup(entry); fill(); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    W128_STAMP_TAKE(); if constexpr (W128_STAMP) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); W128_STAMP_ADD(12);
    while (entry >= 0) {
        auto load_prev_cell = [&]() __attribute__((always_inline)) {
            // previous cell of this tile (w128-tiled layout: 16 x 16-byte loads per lane, 1 KB per wave instruction), issued before the
            // next tile's set-up; the cell update consumes them block by block as they land.  (Issued at the start of the K loop they would
            // be live across it, and hipcc spills all 64 registers -- load, wait, scratch store -- although the loop itself uses 125.)
            if (has_prev) {
    #pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsP, cell_voff + r * 1024, 0, W128_CELL_LD_AUX);
                    cellreg[r] = f32x4_t{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
                }
            } else {
    #pragma unroll
                for (int r = 0; r < 16; ++r) cellreg[r] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            }
        };
        // The tile's first operands have landed: they were issued before the previous tile's 32 result stores (this wave's own
        // pieces; vector memory operations of a wave retire in issue order), so the stores may stay in flight.  Fragments of k-step 0.
        asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        W128_FOR(8, q, { frag_read(w128_c<0>{}, q, w128_c<0>{}, w128_c<0>{}, w128_c<0>{}, w128_c<0>{}); });
        if constexpr (W128_EARLY_CELL) load_prev_cell();
        W128_STAMP_TAKE();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (W128_STAMP) { tfirst += (unsigned)tnow - tlast; tlast = (unsigned)tnow; }

        dy_c = 0; cc_c = 0; first_slab = true;
        for (int j = 0; j < NJ; j += 2) {
            macro_step(w128_c<0>{});
            if (j + 1 < NJ) macro_step(w128_c<1>{});
        }
        // the MFMAs are opaque to hipcc's hazard recognizer: let the last ones retire before the accumulators are read
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
        asm volatile("s_barrier" ::: "memory");        // every wave is done with the operand buffers

        // what the cell update of THIS tile still needs, before set-up overwrites it for the next one
        const __amdgpu_buffer_rsrc_t rsH_t = rsH, rsC_t = rsC;
        int hv_t[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) hv_t[i] = hvoff[i];
        const int cell_voff_t = cell_voff;
        const uint32_t bias_addr_t = bias_addr;

        if constexpr (!W128_EARLY_CELL) load_prev_cell();
        // next tile: set-up, first operands on their way under the cell update
        ++li;
        const int next = __builtin_amdgcn_readfirstlane(llist[li]);
        int has_prev_n = 0;
        if (next >= 0) { has_prev_n = setup(next); fill(); }

        W128_STAMP_TAKE(); if constexpr (W128_STAMP) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tsetup += (unsigned)tnow - tlast; tlast = (unsigned)tnow; }
        // ---- cell update (submodules.py:205-212) straight from the accumulators: lane holds, for pixel (i, p31), rows
        // e = 4 q + gate of gate block j <-> hidden channel j*8 + 2 q + hi
        W128_FOR(NT, jc, {
            constexpr int j = decltype(jc)::value;
            const uint32_t ba = bias_addr_t;
            f32x4_t bq0, bq1, bq2, bq3;                   // pre-scaled biases (in, remember, out, cell) of channels j*8 + 2 q + hi
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bq0) : "v"(ba), "n"((j * 32 + 0) * 4) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bq1) : "v"(ba), "n"((j * 32 + 8) * 4) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bq2) : "v"(ba), "n"((j * 32 + 16) * 4) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bq3) : "v"(ba), "n"((j * 32 + 24) * 4) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq0), "+v"(bq1), "+v"(bq2), "+v"(bq3) :: "memory");
            const f32x4_t bq[4] = {bq0, bq1, bq2, bq3};
            W128_FOR(MT, ic, {
                constexpr int i = decltype(ic)::value;
                float hq[4];
                f32x4_t cr = cellreg[i * 4 + j];
                // the tile stays in its AGPRs up to here: without this, hipcc copies all 256 accumulator registers to VGPRs right behind
                // the K loop (element extraction of 512-bit tuples), in front of the next tile's set-up, and spills
                asm volatile("" : "+a"(acc[i * 4 + j]));
                const f32x16_t tv = acc[i * 4 + j];
                W128_FOR(4, qc, {
                    constexpr int q = decltype(qc)::value;
                    const float gi = tv[q * 4 + 0], gr = tv[q * 4 + 1], go = tv[q * 4 + 2], gc = tv[q * 4 + 3];
                    const float si = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(gi, -L2E, bq[q][0])));
                    const float sr = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(gr, -L2E, bq[q][1])));
                    const float so = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(go, -L2E, bq[q][2])));
                    const float tg = __builtin_fmaf(__builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(gc, 2.0f * L2E, bq[q][3]))), -2.0f, 1.0f);
                    const float nc = __builtin_fmaf(sr, cr[q], si * tg);                              // submodules.py:211
                    const float th = __builtin_fmaf(__builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(nc * (2.0f * L2E))), -2.0f, 1.0f);
                    cr[q] = nc;
                    hq[q] = so * th;                                                                    // submodules.py:212
                });
                cellreg[i * 4 + j] = cr;
                // hidden: the lane holds channels 2 q + hi of the block; after swapping (q0, q2) and (q1, q3) with lane ^ 32 it holds
                // the four consecutive channels 4 hi .. 4 hi + 3 = one 8-byte piece
                float h0 = hq[0], h1 = hq[1], h2 = hq[2], h3 = hq[3];
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(h0), "+v"(h2));
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(h1), "+v"(h3));
                const unsigned w0 = pack_bf16x2(h0, h2), w1 = pack_bf16x2(h1, h3);
                __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{w0, w1}, rsH_t, hv_t[i] + j * 16, 0, W128_H_ST_AUX);
                if constexpr ((i + 1) % W128_E1_BLOCK == 0) __builtin_amdgcn_sched_barrier(0);   // W128_E1_BLOCK (pixel block, gate block) pairs = 4 x that many cells in flight
            });
        });
#pragma unroll
        for (int r = 0; r < 16; ++r)
            __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{__float_as_uint(cellreg[r][0]), __float_as_uint(cellreg[r][1]), __float_as_uint(cellreg[r][2]), __float_as_uint(cellreg[r][3])},
                                                   rsC_t, cell_voff_t + r * 1024, 0, W128_CELL_ST_AUX);
        W128_STAMP_TAKE(); if constexpr (W128_STAMP) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); W128_STAMP_ADD(14);
        entry = next; has_prev = has_prev_n;
    }
#undef W128_RD_P
#undef W128_RD_P0
#undef W128_RD_W
    if constexpr (W128_STAMP) {
        if (lane == 0) {
            float* dbg = const_cast<float*>(reinterpret_cast<const float*>(g.sched)) + (size_t)gridDim.x * g.sched_stride;
            tacc[15] = (unsigned)li; tacc[13] = tfirst; tacc[12] = tsetup;
#pragma unroll
            for (int k = 0; k < 16; ++k) dbg[(blockIdx.x * 4 + wave) * 16 + k] = (float)tacc[k];
        }
    }
}

// w128-tiled cell state <-> NHWC fp32 [M][C] (state import / export, tests): element (m, hc) of the NHWC tensor lives at
// tile (m / 256, hc / 64), wave ((m % 256) / 128) * 2 + (hc % 64) / 32, register r = ((m % 128) / 32) * 4 + (hc % 32) / 8,
// lane (m % 32) + 32 * (hc & 1), component q = (hc % 8) / 2.  Rows m >= M of the tiled buffer are padding (written as zero).
__global__ __launch_bounds__(256) void w128_cell_relayout_kernel(const float* __restrict__ src, float* __restrict__ dst, long long M, int C, int to_tiled) {
    const long long total = ((M + 255) / 256) * 256 * C;
    const int tiles_n = C / 64;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        // idx enumerates the TILED buffer
        const long long tile = idx >> 14;
        const int w = (int)(idx >> 12) & 3, r = (int)(idx >> 8) & 15, lane = (int)(idx >> 2) & 63, q = (int)idx & 3;
        const long long tile_m = tile / tiles_n; const int tile_n = (int)(tile - tile_m * tiles_n);
        const int i = r >> 2, j = r & 3;
        const long long m = tile_m * 256 + (w >> 1) * 128 + i * 32 + (lane & 31);
        const int hc = tile_n * 64 + (w & 1) * 32 + j * 8 + 2 * q + (lane >> 5);
        if (to_tiled) dst[idx] = m < M ? src[m * C + hc] : 0.0f;
        else if (m < M) dst[m * C + hc] = src[idx];
    }
}
