// Event -> voxel-grid builders for gfx950 (K1 tri-linear, K1' nearest-xy, event histogram).
//
// Design (MI355X-first, HBM-bound integer/scatter work -- no MFMA):
//   The reference scatters 8 (tri-linear) or 2 (nearest) read-modify-writes per event into a
//   C x H x W grid.  Global fp32 atomics across the 8 non-coherent XCD L2s would execute
//   memory-side, so the splat is made OUTPUT-STATIONARY and free of global atomics:
//     S  sort     workgroup (segment, slice of 2048 events): the events are loaded ONCE (all column loads of a lane issued
//                 before the first use, then the rectify gathers); count + rank in one pass of returning LDS atomics, LDS
//                 scan, records staged in LDS, then one coalesced non-temporal block copy of the slice's tile-sorted
//                 16-byte records into the slice's FIXED region + a column of the transposed run table (plain stores)
//     D  splat    workgroup (segment, tile), XCD-aware item order: gathers its run from every slice of the segment (run
//                 starts one per lane, DPP wave scan, run lookup by LDS marks), records -> LDS accumulators, then every
//                 output voxel is written ONCE, coalesced (float4 per lane, non-temporal); no pre-zeroing pass.
//   (The nearest-xy path still uses the first pipeline: count -> two scans -> scatter -> splat.  For the tri-linear
//   path it was measured slower -- 0.578 / 0.780 ms (fp32 SoA / raw+rectify) against 0.513 / 0.583 ms -- because it
//   reads the events twice and scatters records into ~450-byte cursor ranges whose boundary lines are shared between
//   workgroups on different XCDs, and has been removed.)
//   What bounds the two kernels was measured with PMC counters and phase ablations: the splat was
//   VALU-issue bound (803 instructions per wave for ~1.5 records per lane) until its corners became straight-line code and
//   its run lookup a mark row; it now sits on its 0.9 GB write (floor 150-160 us) plus the latency of the record fetch.
//   The sort lives on memory-level parallelism (16 waves per CU behind a 36 KB staging buffer).  Design choices that
//   came out of the ablations: 16-byte records that carry the rectified coordinates (a second rectify-map gather in the
//   splat costs more than the bytes it saves), a TRANSPOSED run table (a tile's run starts are one contiguous row) and
//   fixed record regions per slice instead of an allocator.
//   An event whose 2x2 pixel footprint straddles a tile edge is binned into each tile it touches
//   (<= 4); each tile only accumulates the corners it owns.
//
//   Accumulation is FIXED POINT with LDS integer atomics: measured on MI355X, LDS fp32 atomics
//   (ds_add_f32) run ~10x slower than LDS integer atomics (0.40 ms vs 0.04 ms for the 134 M corner
//   updates of one B=8 batch).  A tile whose record count bounds every voxel sum below 2^10 uses 32-bit
//   accumulators at 2^-20 .. 2^-24 (the scale follows the bound, so sparse tiles resolve a float32 ulp): half
//   the LDS (8 workgroups per CU), ds_add_u32, and one v_cvt each way instead of f64 arithmetic.  Denser
//   tiles take the 64-bit path (2^-38, ds_add_u64) in two half-height passes over the same LDS.  Either way
//   the result is the rounded sum of the per-event weights: deterministic and order independent.
//
// Parity: per-event index math and weights follow the reference's float32 / float64 operation
// order exactly (compiled with -ffp-contract=off; IEEE division), so indices are bit-exact and
// every individual contribution is bit-identical; the only difference from the reference is that
// the reference rounds after every sequential fp32 add while this kernel rounds once.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "oess.h"
#include "oess_common.h"

namespace {

typedef float f32x4_t __attribute__((ext_vector_type(4)));
// Streaming store: the voxel grid (0.9 GB per batch) and the sorted records are written once and never re-read by the
// writing kernel; non-temporal stores keep them from allocating in L2 (tools/probes/wg_floor_probe: the grid's tiled write
// pattern runs at 161 us with them, 196 us without).
__device__ __forceinline__ void store_stream(float4* p, const float4 v) {
    __builtin_nontemporal_store(f32x4_t{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4_t*>(p));
}

constexpr int TW = 64;            // tile width (pixels) = one wave of lanes
constexpr int THREADS = 256;
constexpr int EPT = 8;            // events per thread per batch in count / scatter
constexpr int SLICE = THREADS * EPT * 4;        // events per (segment, slice) workgroup = 8192
constexpr int MAX_LDS_TILE_BYTES = 40 * 1024;   // 4 splat workgroups per CU (160 KiB LDS)
constexpr float FIX_SCALE = 274877906944.0f;    // 2^38
constexpr double FIX_INV = 1.0 / 274877906944.0;

struct Geom {
    int C;        // channels accumulated in LDS per tile (tri-linear: bins; nearest: 2*bins)
    int H, W;     // sensor size used for the reference's validity masks
    int Hout;     // rows kept (H - crop_rows)
    int TH;       // tile height (power of two)
    int lgTH;     // log2(TH)
    int tilesX, tilesY, nTiles;
    int nSlices;  // slices per segment (count/scatter workgroups)
};

__host__ Geom make_geom(int C, int H, int W, int crop_rows, int64_t max_seg_len, int lds_budget = MAX_LDS_TILE_BYTES,
                        int acc_bytes = 8) {
    Geom g;
    g.C = C; g.H = H; g.W = W; g.Hout = H - crop_rows;
    int th = lds_budget / (C * TW * acc_bytes);
    if (th > 32) th = 32;
    int lg = 0;
    while ((2 << lg) <= th) ++lg;          // round down to a power of two: tile math is shifts, not divides
    th = 1 << lg;
    g.TH = th; g.lgTH = lg;
    g.tilesX = (W + TW - 1) / TW;
    g.tilesY = (g.Hout + th - 1) / th;
    g.nTiles = g.tilesX * g.tilesY;
    int64_t ns = (max_seg_len + SLICE - 1) / SLICE;
    g.nSlices = (int)(ns < 1 ? 1 : ns);
    return g;
}

// round-to-nearest f32 -> 2^-38 fixed point without the emulated f32->i64 conversion: one f64 fma against
// 1.5*2^52 leaves the integer in the low mantissa bits (|w| < 2^13).
__device__ __forceinline__ long long to_fix(float w) {
    const double t = __fma_rn((double)w, (double)FIX_SCALE, 6755399441055744.0);
    return (__double_as_longlong(t) << 13) >> 13;
}
__device__ __forceinline__ float from_fix(long long a) { return (float)((double)a * FIX_INV); }

// ---------------------------------------------------------------------------------------------
// Event sources.  rec = {x, y, t_norm, value} for the tri-linear splat.
// ---------------------------------------------------------------------------------------------
struct TriRec { float x, y, tn, v; };

struct SrcF32 {                         // VoxelGrid.convert's own arguments
    const float* x; const float* y; const float* p; const float* t;
    int seg_base_index;                 // first segment of the chunk being processed (unused here)
    struct Seg { float t0, denom; };
    __device__ Seg seg(int /*s*/, int64_t b, int64_t e) const {
        Seg sg; sg.t0 = t[b]; sg.denom = __fsub_rn(t[e - 1], sg.t0); return sg;
    }
    using Rec = float4;                 // {x, y, t_norm, value}
    static __device__ Rec sentinel() { return make_float4(0.f, 0.f, 2.0e9f, 0.f); }         // t_norm sentinel: no valid bin
    __device__ Rec pack(const TriRec& r) const { return make_float4(r.x, r.y, r.tn, r.v); }
    __device__ TriRec unpack(const Rec& q) const { TriRec r; r.x = q.x; r.y = q.y; r.tn = q.z; r.v = q.w; return r; }
    // `at(base)`: the same source with its columns advanced to event `base`; load() then takes a 32-bit offset, so the
    // address of every column is a uniform base + a lane offset (one VGPR) instead of a 64-bit pointer per column per event.
    __device__ SrcF32 at(int64_t base) const { SrcF32 q = *this; q.x += base; q.y += base; q.p += base; q.t += base; return q; }
    // The load is split in three so that the sort kernel can issue the column loads of all its events, then the dependent
    // gathers, and only then compute: fetch (column loads), coords (SrcRaw: the rectify-map gather), finish (arithmetic).
    struct Ev { float x, y, t, p; };
    __device__ Ev fetch(unsigned int i) const { Ev v; v.x = x[i]; v.y = y[i]; v.t = t[i]; v.p = p[i]; return v; }
    // 8 CONSECUTIVE events of one lane: two 16-byte loads per column (8 instead of 32 load instructions per lane, and a wave
    // instruction covers 1 KB of a column instead of one 256-byte line)
    __device__ void fetch8(unsigned int i0, Ev (&ev)[8]) const {
        typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
        const float* cols[4] = {x, y, t, p};
        float v[4][8];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f4u a = *reinterpret_cast<const f4u*>(cols[c] + i0), b = *reinterpret_cast<const f4u*>(cols[c] + i0 + 4);
            v[c][0] = a[0]; v[c][1] = a[1]; v[c][2] = a[2]; v[c][3] = a[3]; v[c][4] = b[0]; v[c][5] = b[1]; v[c][6] = b[2]; v[c][7] = b[3];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) { ev[k].x = v[0][k]; ev[k].y = v[1][k]; ev[k].t = v[2][k]; ev[k].p = v[3][k]; }
    }
    __device__ float2 coords(const Ev& v, const Seg&) const { return make_float2(v.x, v.y); }
    template <bool UNIT_DENOM>
    __device__ TriRec finish(const Ev& v, float2 xy, const Seg& sg, int C) const {
        TriRec r;
        r.x = xy.x; r.y = xy.y;
        // representations.py:25  (C-1)*(t-t[0]) / (t[-1]-t[0])   float32, left to right
        const float num = __fmul_rn((float)(C - 1), __fsub_rn(v.t, sg.t0));
        r.tn = UNIT_DENOM ? num : num / sg.denom;                 // x/1 == x exactly: skip the IEEE divide
        r.v = __fsub_rn(__fmul_rn(2.0f, v.p), 1.0f);              // representations.py:31
        return r;
    }
    __device__ bool unit_denom(const Seg& sg) const { return sg.denom == 1.0f; }
};

struct SrcRaw {                         // raw DSEC columns + rectify map (sequence_ov.py:154-157,204-210)
    const uint16_t* x; const uint16_t* y; const int64_t* t; const uint8_t* p;
    const float* maps; const int32_t* seg_map; int H, W;
    int seg_base_index;                 // first segment of the chunk being processed
    struct Seg { int64_t t0; float dlast; float tn0, denom; const float* map; };
    __device__ Seg seg(int s, int64_t b, int64_t e) const {
        Seg sg; sg.t0 = t[b];
        sg.dlast = (float)(double)(t[e - 1] - sg.t0);              // (t-t[0]).astype('float32')[-1]
        float first = 0.0f / sg.dlast;                            // t/t[-1] at index 0 (NaN if dlast==0)
        float last = sg.dlast / sg.dlast;
        sg.tn0 = first; sg.denom = __fsub_rn(last, first);
        sg.map = maps + (size_t)seg_map[seg_base_index + s] * (size_t)H * W * 2;
        return sg;
    }
    using Rec = float4;                 // {x', y', t_norm, value}: the RECTIFIED coordinates travel with the record -- an 8-byte
    //                                     {x | y << 12 | p << 24, t_norm} record + a second map gather in the splat was measured
    //                                     slower (the pipeline is bound by cache-line REQUESTS, not bytes: 17 M more gathers)
    static __device__ Rec sentinel() { return make_float4(0.f, 0.f, 2.0e9f, 0.f); }
    __device__ Rec pack(const TriRec& r) const { return make_float4(r.x, r.y, r.tn, r.v); }
    __device__ TriRec unpack(const Rec& q) const { TriRec r; r.x = q.x; r.y = q.y; r.tn = q.z; r.v = q.w; return r; }
    __device__ SrcRaw at(int64_t base) const { SrcRaw q = *this; q.x += base; q.y += base; q.t += base; q.p += base; return q; }
    struct Ev { int x, y; int64_t t; int p; };
    __device__ Ev fetch(unsigned int i) const { Ev v; v.x = x[i]; v.y = y[i]; v.t = t[i]; v.p = p[i]; return v; }
    // 8 CONSECUTIVE events of one lane: x, y one 16-byte load each, t four, p one 8-byte load (7 instead of 32 load instructions
    // per lane; the columns are only element-aligned, the hardware takes unaligned vector loads)
    __device__ void fetch8(unsigned int i0, Ev (&ev)[8]) const {
        typedef unsigned int u4a2 __attribute__((ext_vector_type(4), aligned(2)));
        typedef unsigned int u4a8 __attribute__((ext_vector_type(4), aligned(8)));
        typedef unsigned int u2a1 __attribute__((ext_vector_type(2), aligned(1)));
        const u4a2 xv = *reinterpret_cast<const u4a2*>(x + i0), yv = *reinterpret_cast<const u4a2*>(y + i0);
        const u2a1 pv = *reinterpret_cast<const u2a1*>(p + i0);
        u4a8 tv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) tv[q] = *reinterpret_cast<const u4a8*>(t + i0 + 2 * q);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            ev[k].x = (int)((xv[k >> 1] >> (16 * (k & 1))) & 0xffffu);
            ev[k].y = (int)((yv[k >> 1] >> (16 * (k & 1))) & 0xffffu);
            ev[k].p = (int)((pv[k >> 2] >> (8 * (k & 3))) & 0xffu);
            const unsigned int lo = tv[k >> 1][2 * (k & 1)], hi = tv[k >> 1][2 * (k & 1) + 1];
            ev[k].t = (int64_t)(((unsigned long long)hi << 32) | lo);
        }
    }
    __device__ float2 coords(const Ev& v, const Seg& sg) const {
        const int xi = v.x >= W ? W - 1 : v.x;                     // reference asserts x.max() < width
        const int yi = v.y >= H ? H - 1 : v.y;
        return *reinterpret_cast<const float2*>(sg.map + (unsigned int)(yi * W + xi) * 2u);
    }
    template <bool UNIT_DENOM>
    __device__ TriRec finish(const Ev& v, float2 xy, const Seg& sg, int C) const {
        TriRec r;
        r.x = xy.x; r.y = xy.y;
        const int64_t d = v.t - sg.t0;
        // int64 -> float64 -> float32 is a single rounding of an exact integer; for |d| < 2^31 the native
        // int32 -> float32 conversion gives the identical result.  Wave-uniform choice: windows are far shorter than 2^31 us.
        float df = (float)(int)d;
        if (__any(d != (int64_t)(int)d)) df = (float)(double)d;
        const float tt = df / sg.dlast;
        const float num = __fmul_rn((float)(C - 1), __fsub_rn(tt, sg.tn0));
        r.tn = UNIT_DENOM ? num : num / sg.denom;
        r.v = __fsub_rn(__fmul_rn(2.0f, (float)v.p), 1.0f);
        return r;
    }
    __device__ bool unit_denom(const Seg& sg) const { return sg.denom == 1.0f; }
};

// Tiles touched by a tri-linear event, as straight-line code: the event's pixel pair (x0, x0+1) x (y0, y0+1) meets at most
// 2 x 2 tiles, tile(a, b) = t00 + a * tilesX + b.  Returns (t00 + tilesX + 1) << 4 | mask with mask bit (2a + b) set when
// tile (a, b) holds a valid corner; the second column / row only counts when it is a different tile from the first.
// Membership is purely spatial; the splat applies the time-bin masks.  NaN coordinates clamp to -8: no tile.
__device__ __forceinline__ unsigned int tri_tiles(float x, float y, const Geom& g) {
    // clamp before the int conversion so that huge coordinates stay "far outside" instead of UB
    const float fx = fminf(fmaxf(x, -8.0f), (float)g.W + 8.0f);
    const float fy = fminf(fmaxf(y, -8.0f), (float)g.H + 8.0f);
    const int x0 = (int)fx, y0 = (int)fy;           // C-style truncation (representations.py:27-28)
    const int cxa = x0 >> 6, cxb = (x0 + 1) >> 6, cya = y0 >> g.lgTH, cyb = (y0 + 1) >> g.lgTH;
    const bool vxa = (unsigned int)x0 < (unsigned int)g.W;
    const bool vxb = (unsigned int)(x0 + 1) < (unsigned int)g.W && (!vxa || cxb != cxa);
    const bool vya = (unsigned int)y0 < (unsigned int)g.Hout;
    const bool vyb = (unsigned int)(y0 + 1) < (unsigned int)g.Hout && (!vya || cyb != cya);
    const unsigned int mask = (unsigned int)(vya && vxa) | (unsigned int)(vya && vxb) << 1 | (unsigned int)(vyb && vxa) << 2 |
                              (unsigned int)(vyb && vxb) << 3;
    return (unsigned int)(cya * g.tilesX + cxa + g.tilesX + 1) << 4 | mask;
}

// ---------------------------------------------------------------------------------------------
// Nearest-xy records (generate_voxel_grid).  rec = {x | y<<16, tis | is_pos<<31, vals_left, vals_right}
// ---------------------------------------------------------------------------------------------
struct NearRec { uint32_t xy; uint32_t tp; float vl, vr; };

template <typename T>
struct SrcNear {
    const T* ev;           // [N x 4] rows (x, y, t, p)
    int nbins;
    struct Seg { T first; double deltaT; };
    __device__ Seg seg(int /*s*/, int64_t b, int64_t e) const {
        Seg sg; sg.first = ev[b * 4 + 2];
        T d = ev[(e - 1) * 4 + 2] - sg.first;                       // data_util.py:66-72
        sg.deltaT = (d == (T)0) ? 1.0 : (double)d;
        return sg;
    }
    // returns false if the event is dropped by the reference's masks
    __device__ bool load(int64_t i, const Seg& sg, const Geom& g, NearRec& r, int& tile) const {
        T ex = ev[i * 4 + 0], ey = ev[i * 4 + 1], et = ev[i * 4 + 2], ep = ev[i * 4 + 3];
        // data_util.py:76  ts = (bins-1) * (t - first) / deltaT : integer product for int64 input
        double ts = (double)((T)(nbins - 1) * (et - sg.first)) / sg.deltaT;
        double exd = (double)ex, eyd = (double)ey;
        tile = 0;
        r.xy = 0; r.tp = 0x7fffffffu; r.vl = 0.f; r.vr = 0.f;
        if (!(exd > -1.0e9 && exd < 1.0e9 && eyd > -1.0e9 && eyd < 1.0e9)) return false;
        long long xs = (long long)ex, ys = (long long)ey;           // astype(int64): truncation
        if (!(ts >= 0.0 && ts < (double)nbins)) return false;       // also rejects NaN
        if (!(xs >= 0 && xs < g.W && ys >= 0 && ys < g.H)) return false;   // valid_pos, data_util.py:88
        if (ys >= g.Hout) return false;                              // cropped rows
        double pol = (double)ep;
        if (pol == 0.0) pol = -1.0;                                  // data_util.py:79
        long long tis = (long long)ts;
        double dts = ts - (double)tis;
        double ap = fabs(pol);
        r.xy = (uint32_t)xs | ((uint32_t)ys << 16);
        r.tp = (uint32_t)tis | ((pol == 1.0) ? 0x80000000u : 0u);
        r.vl = (float)(ap * (1.0 - dts));
        r.vr = (float)(ap * dts);
        tile = ((int)ys >> g.lgTH) * g.tilesX + ((int)xs >> 6);
        return true;
    }
};

// ---------------------------------------------------------------------------------------------
// Nearest-xy pipeline.  Pass A (MODE 0): count     Pass C (MODE 1): scatter.    grid = (nSlices, n_seg)
// table layout: [n_seg][nSlices][nTiles] ints (counts in A; exclusive local offsets after B)
// ---------------------------------------------------------------------------------------------
template <int MODE, typename Src>
__global__ __launch_bounds__(THREADS) void near_bin_kernel(Src src, const int64_t* __restrict__ seg_off, Geom g,
                                                           int* __restrict__ table, const int* __restrict__ seg_base,
                                                           float4* __restrict__ recs, uint32_t cap) {
    extern __shared__ int lds[];
    const int s = blockIdx.y, slice = blockIdx.x;
    const int64_t b = seg_off[s], e = seg_off[s + 1];
    const int64_t n = e - b;
    int* tab = table + ((size_t)s * g.nSlices + slice) * g.nTiles;
    const int64_t sl_beg = (int64_t)slice * SLICE;
    if (MODE == 0) {
        for (int i = threadIdx.x; i < g.nTiles; i += THREADS) lds[i] = 0;
    } else {
        const int base = seg_base[s];
        for (int i = threadIdx.x; i < g.nTiles; i += THREADS) lds[i] = base + tab[i];
    }
    __syncthreads();
    if (sl_beg < n) {
        int64_t sl_end = sl_beg + SLICE;
        if (sl_end > n) sl_end = n;
        const typename Src::Seg sg = src.seg(s, b, e);
        for (int64_t first = sl_beg; first < sl_end; first += THREADS * EPT) {
            NearRec rec[EPT];
            bool ok[EPT];
            int tile_of[EPT];
#pragma unroll
            for (int k = 0; k < EPT; ++k) {
                const int64_t i = first + k * THREADS + threadIdx.x;
                const bool in = i < sl_end;
                ok[k] = src.load(b + (in ? i : sl_end - 1), sg, g, rec[k], tile_of[k]) && in;
            }
#pragma unroll
            for (int k = 0; k < EPT; ++k) {
                if (!ok[k]) continue;
                const int pos = atomicAdd(&lds[tile_of[k]], 1);
                if (MODE == 1 && (uint32_t)pos < cap)
                    recs[pos] = make_float4(__uint_as_float(rec[k].xy), __uint_as_float(rec[k].tp), rec[k].vl, rec[k].vr);
            }
        }
    }
    if (MODE == 0) {
        __syncthreads();
        for (int i = threadIdx.x; i < g.nTiles; i += THREADS) tab[i] = lds[i];
    }
}

// ---------------------------------------------------------------------------------------------
// Pass B1: per segment, exclusive scan of table[s] in (tile, slice) order, in place.
//          tile_start[s][tile] = local start of the tile; seg_total[s] = records of the segment.
// Pass B2: exclusive scan of seg_total -> seg_base (single workgroup).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int block_excl_scan_1024(int v, int* part, int& total) {
    // inclusive Hillis-Steele over 1024 lanes; returns exclusive prefix, total of the block
    part[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int a = (threadIdx.x >= off) ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += a;
        __syncthreads();
    }
    const int incl = part[threadIdx.x];
    total = part[1023];
    __syncthreads();
    return incl - v;
}

__global__ __launch_bounds__(1024) void scan_seg_kernel(int* __restrict__ table, int* __restrict__ tile_start,
                                                        int* __restrict__ seg_total, Geom g) {
    __shared__ int part[1024];
    const int s = blockIdx.x;
    int* tab = table + (size_t)s * g.nSlices * g.nTiles;
    const int n = g.nSlices * g.nTiles;
    int carry = 0;
    for (int base = 0; base < n; base += 1024) {
        const int j = base + threadIdx.x;                 // j = tile * nSlices + slice
        const int tile = j / g.nSlices, slice = j - tile * g.nSlices;
        const int idx = slice * g.nTiles + tile;
        const int v = (j < n) ? tab[idx] : 0;
        int total;
        const int ex = block_excl_scan_1024(v, part, total) + carry;
        if (j < n) {
            tab[idx] = ex;
            if (slice == 0) tile_start[(size_t)s * (g.nTiles + 1) + tile] = ex;
        }
        carry += total;
    }
    if (threadIdx.x == 0) {
        tile_start[(size_t)s * (g.nTiles + 1) + g.nTiles] = carry;
        seg_total[s] = carry;
    }
}

__global__ __launch_bounds__(1024) void scan_base_kernel(const int* __restrict__ seg_total, int* __restrict__ seg_base,
                                                         int n_seg) {
    __shared__ int part[1024];
    int carry = 0;
    for (int base = 0; base < n_seg; base += 1024) {
        const int j = base + threadIdx.x;
        const int v = (j < n_seg) ? seg_total[j] : 0;
        int total;
        const int ex = block_excl_scan_1024(v, part, total) + carry;
        if (j < n_seg) seg_base[j] = ex;
        carry += total;
    }
}

// ---------------------------------------------------------------------------------------------
// Pass D: splat one (segment, tile) into LDS (64-bit fixed point), then write every voxel once.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void write_tile(const long long* acc, float* __restrict__ out, const Geom& g, int s,
                                           int ch_out, int tx, int ty, bool diff_pol) {
    // acc layout [C][TH][TW]; out layout [(s*ch_out + c)][Hout][W]
    const int x_base = tx * TW, y_base = ty * g.TH;
    const size_t plane = (size_t)g.Hout * g.W;
    const int nb = diff_pol ? g.C / 2 : 0;
    if ((g.W & 3) == 0) {
        const int q_per_row = TW / 4;
        const int total = ch_out * g.TH * q_per_row;
        for (int i = threadIdx.x; i < total; i += THREADS) {
            const int q = i % q_per_row, rc = i / q_per_row;
            const int rr = rc & (g.TH - 1), c = rc >> g.lgTH;
            const int xx = x_base + q * 4, yy = y_base + rr;
            if (xx < g.W && yy < g.Hout) {
                const long long* a = &acc[(c * g.TH + rr) * TW + q * 4];
                float4 v;
                if (diff_pol) {                      // voxel_grid_positive - voxel_grid_negative (data_util.py:116)
                    const long long* m = &acc[((c + nb) * g.TH + rr) * TW + q * 4];
                    v = make_float4(from_fix(a[0]) - from_fix(m[0]), from_fix(a[1]) - from_fix(m[1]),
                                    from_fix(a[2]) - from_fix(m[2]), from_fix(a[3]) - from_fix(m[3]));
                } else {
                    v = make_float4(from_fix(a[0]), from_fix(a[1]), from_fix(a[2]), from_fix(a[3]));
                }
                *reinterpret_cast<float4*>(&out[((size_t)s * ch_out + c) * plane + (size_t)yy * g.W + xx]) = v;
            }
        }
    } else {
        const int total = ch_out * g.TH * TW;
        for (int i = threadIdx.x; i < total; i += THREADS) {
            const int q = i % TW, rc = i / TW;
            const int rr = rc & (g.TH - 1), c = rc >> g.lgTH;
            const int xx = x_base + q, yy = y_base + rr;
            if (xx < g.W && yy < g.Hout) {
                float v = from_fix(acc[(c * g.TH + rr) * TW + q]);
                if (diff_pol) v -= from_fix(acc[((c + nb) * g.TH + rr) * TW + q]);
                out[((size_t)s * ch_out + c) * plane + (size_t)yy * g.W + xx] = v;
            }
        }
    }
}

__device__ __forceinline__ void lds_add(long long* p, long long v) {
    atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v);       // ds_add_u64
}

constexpr int PRE = 3;     // record loads issued before the LDS zero fill

__global__ __launch_bounds__(THREADS) void near_splat_kernel(const float4* __restrict__ recs,
                                                             const int* __restrict__ tile_start,
                                                             const int* __restrict__ seg_base, Geom g, int nbins,
                                                             int separate_pol, int count_mode, uint32_t cap,
                                                             float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) long long acc[];   // [2*nbins][TH][TW]: pos bins then neg bins
    const int tile = blockIdx.x, s = blockIdx.y;
    const int ty = tile / g.tilesX, tx = tile - ty * g.tilesX;
    const int lds_n = g.C * g.TH * TW;
    const int base = seg_base[s];
    const uint32_t beg = (uint32_t)(base + tile_start[(size_t)s * (g.nTiles + 1) + tile]);
    uint32_t end = (uint32_t)(base + tile_start[(size_t)s * (g.nTiles + 1) + tile + 1]);
    if (end > cap) end = cap;
    float4 pre[PRE];
#pragma unroll
    for (int k = 0; k < PRE; ++k) {
        const uint32_t i = beg + k * THREADS + threadIdx.x;
        // sentinel: tis = 0x7fffffff -> neither bin test passes
        pre[k] = (i < end) ? recs[i] : make_float4(0.f, __uint_as_float(0x7fffffffu), 0.f, 0.f);
    }
    for (int i = threadIdx.x; i < lds_n; i += THREADS) acc[i] = 0;
    __syncthreads();
    const int x_lo = tx * TW, y_lo = ty * g.TH;
    auto splat = [&](const float4 r) {
        const uint32_t xy = __float_as_uint(r.x), tp = __float_as_uint(r.y);
        const int lx = (int)(xy & 0xffffu) - x_lo, ly = (int)(xy >> 16) - y_lo;
        const uint32_t tis = tp & 0x7fffffffu;
        const int pol_base = (tp & 0x80000000u) ? 0 : nbins;
        float vl = r.z, vr = r.w;
        if (count_mode) { vl = 1.0f; vr = 1.0f; }
        if (tis < (uint32_t)nbins)                                                   // data_util.py:86-93
            lds_add(&acc[((pol_base + (int)tis) * g.TH + ly) * TW + lx], to_fix(vl));
        if (tis + 1u < (uint32_t)nbins)                                              // data_util.py:95-98
            lds_add(&acc[((pol_base + (int)tis + 1) * g.TH + ly) * TW + lx], to_fix(vr));
    };
#pragma unroll
    for (int k = 0; k < PRE; ++k) splat(pre[k]);
    for (uint32_t i = beg + PRE * THREADS + threadIdx.x; i < end; i += THREADS) splat(recs[i]);
    __syncthreads();
    write_tile(acc, out, g, s, separate_pol ? 2 * nbins : nbins, tx, ty, !separate_pol);
}

// =============================================================================================
// Tri-linear pipeline: SLICE-LOCAL SORT + multi-run splat (2 kernels).
//   S  sort   workgroup (segment, slice of 2048 events): load the events ONCE, count + rank per tile with returning
//             LDS atomics, LDS scan, records into an LDS buffer, then ONE coalesced block copy of the slice's records
//             (sorted by tile) to the slice's fixed region; the per-slice table column {start[tile], ..., end,
//             max |value|} is stored with plain stores.
//   D  splat  workgroup (segment, tile): gathers its run from every slice of the segment (run starts one per
//             lane, wave scan, mark-row lookup), accumulates in LDS and writes every voxel once.
// =============================================================================================
constexpr int SORT_THREADS = 256;        // 4 sort workgroups per CU (36 KB staging each); 512 threads / 4096-event slices: 1.5-3 % slower
constexpr int SSL = 2048;                 // events per sort slice
constexpr int SEPT = SSL / SORT_THREADS; // events per thread
constexpr int LCAP = 2304;               // records staged in LDS (36 KB); a slice needs 2048 * ~1.08 on real data,
                                         // up to 4 * 2048 on adversarial input (overflow goes straight to HBM)
constexpr int FAST_LDS_BYTES = 20 * 1024;   // 32-bit accumulator tile: 8 splat workgroups per CU

// Barrier for LDS-only hand-offs: waits for this wave's LDS operations, not for its global loads and stores.  __syncthreads()
// carries a workgroup-scope release, which on gfx9 means s_waitcnt vmcnt(0): it would drain the table / record stores of the
// sort kernel (and any prefetched columns) at every phase boundary.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ int block_incl_scan_256(int v, int* wsum) {
    // inclusive scan over all threads of the block (<= 16 waves); wsum: 16 ints of LDS
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int y = __shfl_up(x, off, 64);
        if (lane >= off) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    lds_barrier();
    int add = 0;
    for (int w = 0; w < wave; ++w) add += wsum[w];
    lds_barrier();
    return x + add;
}

// Run table, TRANSPOSED: T[segment][i][slice], i = 0 .. nTiles: start of tile i's run inside the slice's record region
// (i = nTiles: end of the last run), i = nTiles + 1: float bits of max |value| over the slice's events.  A splat workgroup
// reads rows `tile` and `tile + 1`: two contiguous rows of nSl ints.  Record region of (segment, slice) =
// [(segment * nSl + slice) * RSTRIDE, + RSTRIDE): fixed, no allocator (the workspace holds 4 records per event anyway).
constexpr int RSTRIDE = 4 * SSL;      // worst case: every event of the slice in 4 tiles
template <typename Src>
__global__ __launch_bounds__(SORT_THREADS) void tri_sort_kernel(Src src, const int64_t* __restrict__ seg_off, Geom g, int nSl,
                                                           int* __restrict__ table,
                                                           typename Src::Rec* __restrict__ recs, unsigned int cap) {
    using Rec = typename Src::Rec;
    extern __shared__ __attribute__((aligned(16))) unsigned char sm_sort[];
    const int nT = g.nTiles;
    int* cur = reinterpret_cast<int*>(sm_sort);                                   // [nT + 1]
    int* wsum = cur + nT + 1;                                                     // [16] + max |value|
    Rec* buf = reinterpret_cast<Rec*>(sm_sort + (((size_t)(nT + 1 + 17) * 4 + 15) & ~(size_t)15));   // [LCAP]
    const int s = blockIdx.y, slice = blockIdx.x;
    const int64_t b = seg_off[s], e = seg_off[s + 1];
    const int64_t n = e - b;
    int* tab = table + (size_t)s * (nT + 2) * nSl + slice;          // column `slice` of the segment's transposed table
    const int64_t sl_beg = (int64_t)slice * SSL;
    for (int i = threadIdx.x; i <= nT; i += SORT_THREADS) cur[i] = 0;
    if (threadIdx.x == 0) wsum[16] = 0;
    if (sl_beg >= n) {                                   // empty slice (ragged segments): all-zero column
        for (int i = threadIdx.x; i < nT + 2; i += SORT_THREADS) tab[(size_t)i * nSl] = 0;
        return;
    }
    lds_barrier();
    int64_t sl_end = sl_beg + SSL;
    if (sl_end > n) sl_end = n;
    const typename Src::Seg sg = src.seg(s, b, e);
    const Src here = src.at(b + sl_beg);                 // columns at the slice's first event: 32-bit lane offsets from here on
    const unsigned int n_here = (unsigned int)(sl_end - sl_beg);
    Rec packed[SEPT];
    unsigned int tl[SEPT];                                // tri_tiles code of the event; 0 = no tile (or past the slice's end)
    float vm = 0.f;
    // the only read of the events: SEPT independent column loads per lane, all issued before the first use, then the
    // dependent gathers likewise (the kernel lives on memory-level parallelism; a loop that loads and computes per event
    // measured 10 % slower)
    typename Src::Ev ev[SEPT];
    float2 xy[SEPT];
    // lane L owns events 8 L .. 8 L + 7 of the slice (the order of events inside a slice is irrelevant: ranks come from atomics
    // and the accumulation is order independent); a lane whose 8 events are not all inside the slice takes clamped scalar loads
    static_assert(SEPT == 8, "fetch8");
    const unsigned int e0 = threadIdx.x * SEPT;
    if (e0 + SEPT <= n_here) here.fetch8(e0, ev);
    else {
#pragma unroll
        for (int k = 0; k < SEPT; ++k) ev[k] = here.fetch(min(e0 + k, n_here - 1));
    }
#pragma unroll
    for (int k = 0; k < SEPT; ++k) xy[k] = here.coords(ev[k], sg);
    auto load_all = [&](auto unit_c) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < SEPT; ++k) {
            const bool ok = e0 + k < n_here;
            const TriRec r = here.template finish<decltype(unit_c)::value>(ev[k], xy[k], sg, g.C);
            packed[k] = here.pack(r);
            float av = fabsf(r.v);
            av = (av != av) ? __int_as_float(0x7f800000) : av;                        // NaN value: force the 64-bit path
            vm = fmaxf(vm, ok ? av : 0.f);
            tl[k] = ok ? tri_tiles(r.x, r.y, g) : 0u;
        }
    };
    if (here.unit_denom(sg)) load_all(std::true_type{}); else load_all(std::false_type{});     // block-uniform
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) vm = fmaxf(vm, __shfl_xor(vm, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(&wsum[16], __float_as_int(vm));        // non-negative floats order like ints
    // phase A: count AND rank in one pass of returning LDS atomics (counter of tile t at cur[t + 1], so that the inclusive
    // scan below yields exclusive starts); the rank of a record inside its tile stays in registers, two 16-bit ranks per VGPR
    const int tx1 = g.tilesX + 1;
    unsigned int rk[SEPT][2];
#pragma unroll
    for (int k = 0; k < SEPT; ++k) {
        int* c = &cur[(int)(tl[k] >> 4) - tx1 + 1];      // counter of tile (0, 0) + 1
        unsigned int r0 = 0, r1 = 0, r2 = 0, r3 = 0;
        if (tl[k] & 1u) r0 = (unsigned int)atomicAdd(c, 1);
        if (tl[k] & 2u) r1 = (unsigned int)atomicAdd(c + 1, 1);
        if (tl[k] & 4u) r2 = (unsigned int)atomicAdd(c + g.tilesX, 1);
        if (tl[k] & 8u) r3 = (unsigned int)atomicAdd(c + g.tilesX + 1, 1);
        rk[k][0] = r0 | r1 << 16; rk[k][1] = r2 | r3 << 16;            // a slice holds at most 4 x 2048 records
    }
    lds_barrier();
    const int vmax_bits = wsum[16];
    {   // inclusive scan of cur[0 .. nT] in place
        const int per = (nT + 1 + SORT_THREADS - 1) / SORT_THREADS;
        const int lo = threadIdx.x * per;
        int sum = 0;
        for (int i = lo; i < lo + per && i <= nT; ++i) sum += cur[i];
        const int incl = block_incl_scan_256(sum, wsum);
        int run = incl - sum;
        for (int i = lo; i < lo + per && i <= nT; ++i) { run += cur[i]; cur[i] = run; }
    }
    lds_barrier();
    const int total = cur[nT];
    const unsigned int base = (unsigned int)(((size_t)s * nSl + slice) * RSTRIDE);
    for (int i = threadIdx.x; i <= nT; i += SORT_THREADS) tab[(size_t)i * nSl] = cur[i];                  // starts inside the region
    if (threadIdx.x == 0) tab[(size_t)(nT + 1) * nSl] = vmax_bits;
    Rec* region = recs + base;
    // phase B: position = start of the tile + rank; stage in LDS (the counters are only read from here on)
    auto place = [&](const int* c, unsigned int rank, const Rec& q) __attribute__((always_inline)) {
        const int pos = *c + (int)rank;
        if (pos < LCAP) buf[pos] = q; else if (base + (unsigned int)pos < cap) region[pos] = q;
    };
#pragma unroll
    for (int k = 0; k < SEPT; ++k) {
        const int* c = &cur[(int)(tl[k] >> 4) - tx1];
        if (tl[k] & 1u) place(c, rk[k][0] & 0xffffu, packed[k]);
        if (tl[k] & 2u) place(c + 1, rk[k][0] >> 16, packed[k]);
        if (tl[k] & 4u) place(c + g.tilesX, rk[k][1] & 0xffffu, packed[k]);
        if (tl[k] & 8u) place(c + g.tilesX + 1, rk[k][1] >> 16, packed[k]);
    }
    lds_barrier();
    const int staged = total < LCAP ? total : LCAP;
    for (int i = threadIdx.x; i < staged; i += SORT_THREADS)
        if (base + (unsigned int)i < cap) store_stream(&region[i], buf[i]);        // whole, exclusively owned lines
}

// Wave64 inclusive scan / max on the DPP network (row_shr 1,2,4,8 inside the 16-lane rows, then row_bcast:15 and :31 across
// them): six VALU instructions each and no LDS traffic, where __shfl_up costs a ds_bpermute per step.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_or_zero(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false); }
__device__ __forceinline__ int wave_incl_scan_add(int v) {
    v += dpp_or_zero<0x111, 0xf>(v);
    v += dpp_or_zero<0x112, 0xf>(v);
    v += dpp_or_zero<0x114, 0xf>(v);
    v += dpp_or_zero<0x118, 0xf>(v);
    v += dpp_or_zero<0x142, 0xa>(v);
    v += dpp_or_zero<0x143, 0xc>(v);
    return v;
}
// max over the wave of non-negative floats (0 is the identity), returned wave-uniform
__device__ __forceinline__ float wave_max_nonneg(float f) {
    auto step = [](float a, int b) { return fmaxf(a, __int_as_float(b)); };
    f = step(f, dpp_or_zero<0x111, 0xf>(__float_as_int(f)));
    f = step(f, dpp_or_zero<0x112, 0xf>(__float_as_int(f)));
    f = step(f, dpp_or_zero<0x114, 0xf>(__float_as_int(f)));
    f = step(f, dpp_or_zero<0x118, 0xf>(__float_as_int(f)));
    f = step(f, dpp_or_zero<0x142, 0xa>(__float_as_int(f)));
    f = step(f, dpp_or_zero<0x143, 0xc>(__float_as_int(f)));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(f), 63));
}

constexpr int RUN_CHUNK = 64;            // slices whose runs are gathered per splat iteration: one per lane, kept in
                                         // registers: no LDS beyond the accumulators

// rows [row_lo, row_lo + rows) of tile (tx, .) -> out, every voxel once, float4 per lane
template <typename ACC>
__device__ __forceinline__ void write_rows(const ACC* acc, float inv_scale, float* __restrict__ out, const Geom& g, int s, int tx,
                                           int row_lo, int rows) {
    const int lgr = 31 - __clz(rows);                 // rows is a power of two (TH or TH / 2)
    const int x_base = tx * TW;
    const size_t plane = (size_t)g.Hout * g.W;
    auto cvt = [&](ACC a) -> float {
        if (sizeof(ACC) == 8) return from_fix((long long)a);
        return __fmul_rn((float)(int)a, inv_scale);
    };
    constexpr int Q = TW / 4, RPI = THREADS / Q;      // float4s per tile row, tile rows covered per iteration (16)
    if ((g.W & 3) == 0 && rows <= RPI) {
        // the common shape: a lane keeps its (row, column) and walks the channels with constant strides
        const int q = threadIdx.x & (Q - 1), rc0 = threadIdx.x / Q;
        const int rr = rc0 & (rows - 1), cstep = RPI >> lgr;
        const int xx = x_base + q * 4, yy = row_lo + rr;
        if (xx < g.W && yy < g.Hout) {
            int c = rc0 >> lgr;
            const ACC* a = &acc[(c * rows + rr) * TW + q * 4];
            float* o = &out[((size_t)s * g.C + c) * plane + (size_t)yy * g.W + xx];
            const size_t ostep = (size_t)cstep * plane;
            for (; c < g.C; c += cstep, a += RPI * TW, o += ostep)
                store_stream(reinterpret_cast<float4*>(o), make_float4(cvt(a[0]), cvt(a[1]), cvt(a[2]), cvt(a[3])));
        }
    } else if ((g.W & 3) == 0) {
        const int total = g.C * rows * Q;
        for (int i = threadIdx.x; i < total; i += THREADS) {
            const int q = i & (Q - 1), rc = i / Q;
            const int rr = rc & (rows - 1), c = rc >> lgr;
            const int xx = x_base + q * 4, yy = row_lo + rr;
            if (xx < g.W && yy < g.Hout) {
                const ACC* a = &acc[(c * rows + rr) * TW + q * 4];
                store_stream(reinterpret_cast<float4*>(&out[((size_t)s * g.C + c) * plane + (size_t)yy * g.W + xx]),
                             make_float4(cvt(a[0]), cvt(a[1]), cvt(a[2]), cvt(a[3])));
            }
        }
    } else {
        const int total = g.C * rows * TW;
        for (int i = threadIdx.x; i < total; i += THREADS) {
            const int q = i & (TW - 1), rc = i >> 6;
            const int rr = rc & (rows - 1), c = rc >> lgr;
            const int xx = x_base + q, yy = row_lo + rr;
            if (xx < g.W && yy < g.Hout) out[((size_t)s * g.C + c) * plane + (size_t)yy * g.W + xx] = cvt(acc[(c * rows + rr) * TW + q]);
        }
    }
}

// One record into the LDS tile: the eight corners as straight-line code.  Validity is one unsigned compare per axis end
// (tw_eff / rows_eff already hold the image border), the six axis weights are computed once, and the power-of-two scale of
// the 32-bit accumulators is folded into the value up front (scaling by 2^sh commutes with every rounding of the product
// chain; where it does not - an intermediate below 2^-126 - both orders round to the integer 0).
// representations.py:39  value * (1-|xlim-x|) * (1-|ylim-y|) * (1-|tlim-t_norm|), float32, in that order.
template <bool FAST, bool COUNT>
__device__ __forceinline__ void splat_record(const TriRec r, const Geom& g, int x_lo, unsigned int tw_eff, int row_lo,
                                             unsigned int rows_eff, int rows, float scale, int* acc32) {
    const float x = r.x, y = r.y, tn = r.tn;
    const float fx = fminf(fmaxf(x, -8.0f), (float)g.W + 8.0f);
    const float fy = fminf(fmaxf(y, -8.0f), (float)g.H + 8.0f);
    // NaN/inf time: Tensor.int() gives INT_MIN on the CPU -> every corner masked
    const int x0 = (int)fx, y0 = (int)fy, t0 = (fabsf(tn) < 1.0e9f) ? (int)tn : 0x40000000;
    const int lx = x0 - x_lo, ly = y0 - row_lo;
    const bool vx0 = (unsigned int)lx < tw_eff, vx1 = (unsigned int)(lx + 1) < tw_eff;
    const bool vy0 = (unsigned int)ly < rows_eff, vy1 = (unsigned int)(ly + 1) < rows_eff;
    const bool vt0 = (unsigned int)t0 < (unsigned int)g.C, vt1 = (unsigned int)(t0 + 1) < (unsigned int)g.C;
    const float val = FAST ? __fmul_rn(r.v, scale) : r.v;
    const float wx0 = __fmul_rn(val, __fsub_rn(1.0f, fabsf(__fsub_rn((float)x0, x))));
    const float wx1 = __fmul_rn(val, __fsub_rn(1.0f, fabsf(__fsub_rn((float)(x0 + 1), x))));
    const float by0 = __fsub_rn(1.0f, fabsf(__fsub_rn((float)y0, y)));
    const float by1 = __fsub_rn(1.0f, fabsf(__fsub_rn((float)(y0 + 1), y)));
    const float ct0 = __fsub_rn(1.0f, fabsf(__fsub_rn((float)t0, tn)));
    const float ct1 = __fsub_rn(1.0f, fabsf(__fsub_rn((float)(t0 + 1), tn)));
    const int base = (t0 * rows + ly) * TW + lx;
    const int tstep = rows * TW;
    auto add = [&](bool m, int off, float wxy, float ct) {
        if (!m) return;
        if (FAST) {
            const int wi = COUNT ? (int)scale : __float2int_rn(__fmul_rn(wxy, ct));
            atomicAdd(&acc32[base + off], wi);                                                   // ds_add_u32
        } else {
            lds_add(reinterpret_cast<long long*>(acc32) + (base + off), to_fix(COUNT ? 1.0f : __fmul_rn(wxy, ct)));
        }
    };
    const float w00 = __fmul_rn(wx0, by0), w01 = __fmul_rn(wx0, by1), w10 = __fmul_rn(wx1, by0), w11 = __fmul_rn(wx1, by1);
    add(vx0 && vy0 && vt0, 0, w00, ct0);
    add(vx0 && vy0 && vt1, tstep, w00, ct1);
    add(vx0 && vy1 && vt0, TW, w01, ct0);
    add(vx0 && vy1 && vt1, TW + tstep, w01, ct1);
    add(vx1 && vy0 && vt0, 1, w10, ct0);
    add(vx1 && vy0 && vt1, 1 + tstep, w10, ct1);
    add(vx1 && vy1 && vt0, TW + 1, w11, ct0);
    add(vx1 && vy1 && vt1, TW + 1 + tstep, w11, ct1);
}

// One (segment, tile) item per workgroup.  A persistent grid (8 workgroups per CU walking
// the items, next item's table rows prefetched) was built and measured SLOWER (329 vs 262 us): the hardware dispatcher
// balances the uneven items better than a static loop.
//
// The kernel is VALU-issue bound (r02 PMC: SQ_INSTS_VALU x 4 cycles / 1024 SIMDs = its whole duration), so the code below
// is arranged for instruction count: straight-line corners (splat_record), run lookup by LDS marks instead of a per-lane
// binary search, constant-stride write-out, no integer divisions.
template <typename Src>
__global__ __launch_bounds__(THREADS) void tri_splat_kernel(Src src, const typename Src::Rec* __restrict__ recs,
                                                            const int* __restrict__ table, Geom g, int nSl, int count_mode,
                                                            unsigned int cap, int n_items, float* __restrict__ out) {
    using Rec = typename Src::Rec;
    extern __shared__ __attribute__((aligned(16))) int acc32[];          // [C][TH][TW] ints == [C][TH/2][TW] long longs
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nT = g.nTiles;
    // XCD-aware item order: workgroup L runs on XCD L % 8; giving every XCD one contiguous eighth of the (segment, tile) items
    // puts the neighbouring tiles of a slice - whose runs share their boundary cache lines - on the same L2, one after the other
    const int per_xcd = (n_items + 7) >> 3;
    const int item = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (item >= n_items) return;
    const int s = item / nT, tile = item - s * nT;
    const int ty = tile / g.tilesX, tx = tile - ty * g.tilesX;
    const int x_lo = tx * TW, y_lo = ty * g.TH;
    const int* trow = table + ((size_t)s * (nT + 2) + tile) * nSl;       // run starts of this tile, one per slice; ends follow
    const int* vrow = table + ((size_t)s * (nT + 2) + nT + 1) * nSl;     // max |value| of each slice

    // chunk 0 of the run table: one slice per lane
    int cnt0 = 0; unsigned int beg0 = 0; float vmax = 0.f;
    if (lane < nSl) {
        const int st = trow[lane], en = trow[nSl + lane];
        cnt0 = en - st;
        beg0 = (unsigned int)(((size_t)s * nSl + lane) * RSTRIDE) + (unsigned int)st;
        vmax = __int_as_float(vrow[lane]);
    }
    const int incl0 = wave_incl_scan_add(cnt0);
    const int total0 = __builtin_amdgcn_readlane(incl0, 63);               // SGPR: the loops below are wave-uniform
    // ---- accumulator choice: every voxel sum of this tile is bounded by (records of the tile) x max |value|
    int total_all = total0;
    if (nSl > RUN_CHUNK) {                                               // segments longer than 64 slices (rare)
        int extra = 0;
        for (int c0 = RUN_CHUNK; c0 < nSl; c0 += RUN_CHUNK)
            if (c0 + lane < nSl) {
                extra += trow[nSl + c0 + lane] - trow[c0 + lane];
                vmax = fmaxf(vmax, __int_as_float(vrow[c0 + lane]));
            }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) extra += __shfl_xor(extra, off, 64);
        total_all += __builtin_amdgcn_readfirstlane(extra);
    }
    vmax = wave_max_nonneg(vmax);
    if (count_mode) vmax = 1.0f;
    const float boundf = (float)total_all * vmax + 1.0f;                 // +inf for a NaN / inf value: 64-bit path
    int sh = 0;
    if (boundf < 1024.0f) {
        const int bnd = (int)boundf;                                     // >= 1
        const int lg = (bnd <= 1) ? 0 : 32 - __clz(bnd - 1);             // ceil(log2(bnd))
        sh = 30 - (lg < 1 ? 1 : lg);
        if (sh > 24) sh = 24;
    }
    const bool fast = sh >= 20;                                          // wave- and block-uniform
    const float scale = __int_as_float((127 + sh) << 23), inv_scale = __int_as_float((127 - sh) << 23);
    const unsigned int tw_eff = (unsigned int)min(TW, g.W - x_lo);

    auto run_pass = [&](auto fast_c, auto count_c, const int row_lo, const int rows) __attribute__((always_inline)) {
        constexpr bool FAST = decltype(fast_c)::value, COUNT = decltype(count_c)::value;
        const int lds_n = g.C * rows * TW * (FAST ? 1 : 2);              // in ints
        const unsigned int rows_eff = (unsigned int)max(0, min(rows, g.Hout - row_lo));
        auto splat = [&](const Rec q) __attribute__((always_inline)) {
            splat_record<FAST, COUNT>(src.unpack(q), g, x_lo, tw_eff, row_lo, rows_eff, rows, scale, acc32);
        };
        for (int c0 = 0; c0 < nSl; c0 += RUN_CHUNK) {
            const int nc = (nSl - c0 < RUN_CHUNK) ? nSl - c0 : RUN_CHUNK;
            int cnt = cnt0, incl = incl0, total = total0;
            unsigned int beg = beg0;
            if (c0 > 0) {
                cnt = 0; beg = 0;
                if (lane < nc) {
                    const int st = trow[c0 + lane], en = trow[nSl + c0 + lane];
                    cnt = en - st;
                    beg = (unsigned int)(((size_t)s * nSl + c0 + lane) * RSTRIDE) + (unsigned int)st;
                }
                incl = wave_incl_scan_add(cnt);
                total = __builtin_amdgcn_readlane(incl, 63);
            }
            const int excl = incl - cnt;
            // record j of the concatenated runs, general form: per-lane binary search over the run starts (ds_bpermute).
            // All lanes take part in the shuffles (fixed trip count, j clamped); only the load is predicated.
            auto fetch_search = [&](int j) __attribute__((always_inline)) -> Rec {
                const bool valid = j < total;
                const int jj = valid ? j : 0;
                int lo = 0, hi = nc;                                       // largest run index with excl[idx] <= jj
#pragma unroll
                for (int step = 0; step < 6; ++step) {
                    const int mid = (lo + hi) >> 1;
                    const int v = __shfl(excl, mid, 64);
                    if (hi - lo > 1) { if (v <= jj) lo = mid; else hi = mid; }
                }
                const unsigned int ri = (unsigned int)__shfl((int)beg, lo, 64) + (unsigned int)(jj - __shfl(excl, lo, 64));
                return (valid && ri < cap) ? recs[ri] : Src::sentinel();
            };
            Rec pre[PRE];
            if (c0 == 0) {
                // The first PRE x THREADS records (all of them, for all but the densest tiles) are located without a search
                // while the accumulator LDS is still free.  The non-empty runs are compacted to the low lanes (one bijective
                // ds_permute), so their starts are strictly increasing; a wave whose 64 lanes want records j0 .. j0+63 lets
                // every run starting inside (j0, j0+64) mark its start in a 64-entry LDS row, and lane L's run is
                //   (runs starting at or before j0) - 1 + (marks at positions <= L).
                // The row of wave w lies in words the same wave zero-fills afterwards: no barrier is involved.
                const unsigned long long ne = __ballot(cnt > 0);
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(ne >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)ne, 0u));
                const int nne = __popcll(ne);
                const int dst = (cnt > 0) ? rank : nne + (lane - rank);
                int c_excl = __builtin_amdgcn_ds_permute(dst << 2, excl);
                const int c_off = __builtin_amdgcn_ds_permute(dst << 2, (int)beg - excl);     // record index = c_off[run] + j
                if (lane >= nne) c_excl = 0x7fffffff;
                int* mark = acc32 + wave * 128;                            // words [128 w, 128 w + 64)
#pragma unroll
                for (int k = 0; k < PRE; ++k) {                            // in flight under the LDS zero fill
                    pre[k] = Src::sentinel();
                    if (k * THREADS < total) {
                        const int j0 = k * THREADS + wave * 64;
                        const int d = c_excl - j0;
                        const int below = __popcll(__ballot(d <= 0));
                        mark[lane] = 0;
                        __builtin_amdgcn_wave_barrier();
                        if ((unsigned int)(d - 1) < 63u) mark[d] = 1;
                        __builtin_amdgcn_wave_barrier();
                        const int flag = mark[lane];
                        __builtin_amdgcn_wave_barrier();
                        const unsigned long long m = __ballot(flag != 0);
                        const int run = below - 1 + flag +
                            (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
                        const int j = j0 + lane;
                        const unsigned int ri = (unsigned int)__shfl(c_off, run, 64) + (unsigned int)j;
                        if (j < total && ri < cap) pre[k] = recs[ri];
                    }
                }
                for (int i = threadIdx.x * 2; i < lds_n; i += THREADS * 2) *reinterpret_cast<int2*>(&acc32[i]) = make_int2(0, 0);
                lds_barrier();                             // LDS only: the record loads stay in flight
            } else {
#pragma unroll
                for (int k = 0; k < PRE; ++k)
                    pre[k] = (k * THREADS < total) ? fetch_search(k * THREADS + (int)threadIdx.x) : Src::sentinel();
            }
#pragma unroll
            for (int k = 0; k < PRE; ++k)
                if (k * THREADS < total) splat(pre[k]);
            for (int j0 = PRE * THREADS; j0 < total; j0 += THREADS) splat(fetch_search(j0 + (int)threadIdx.x));
        }
        lds_barrier();
        if (FAST) write_rows<int>(acc32, inv_scale, out, g, s, tx, row_lo, rows);
        else write_rows<long long>(reinterpret_cast<const long long*>(acc32), 0.f, out, g, s, tx, row_lo, rows);
        lds_barrier();     // the accumulators are re-zeroed by the next pass; the grid stores are NOT waited for (a
                           // __syncthreads() here held the workgroup's LDS and wave slots until they were acknowledged)
    };
    auto run_tile = [&](auto count_c) __attribute__((always_inline)) {
        if (fast) {
            run_pass(std::true_type{}, count_c, y_lo, g.TH);
        } else {                                                    // dense tile: two half-height passes, same LDS bytes
            const int half = g.TH >> 1;
            run_pass(std::false_type{}, count_c, y_lo, half);
            run_pass(std::false_type{}, count_c, y_lo + half, half);
        }
    };
    if (count_mode) run_tile(std::true_type{}); else run_tile(std::false_type{});
}

// Event histogram (a4): tiny; direct global atomics on a zeroed 2 x H x W image per segment.
__global__ __launch_bounds__(THREADS) void hist_kernel(const int64_t* __restrict__ ev, const int64_t* __restrict__ seg_off,
                                                       int H, int W, float* __restrict__ out) {
    const int s = blockIdx.y;
    const int64_t b = seg_off[s], e = seg_off[s + 1];
    float* o = out + (size_t)s * 2 * H * W;
    for (int64_t i = b + blockIdx.x * THREADS + threadIdx.x; i < e; i += (int64_t)gridDim.x * THREADS) {
        int64_t x = ev[i * 4 + 0], y = ev[i * 4 + 1], p = ev[i * 4 + 3];
        if (p == 0) p = -1;
        if (x < 0 || x >= W || y < 0 || y >= H) continue;
        if (p == 1) atomicAdd(&o[(size_t)H * W + y * W + x], 1.0f);       // channel 1 = pos
        else if (p == -1) atomicAdd(&o[y * W + x], 1.0f);                  // channel 0 = neg
    }
}

struct Workspace {
    int* table; int* tile_start; int* seg_total; int* seg_base; float4* recs; uint32_t cap;
};

// table sizes depend on nSlices (from max_seg_len); the query uses the worst case n_events per segment.
size_t ws_layout(int64_t n_events, int n_seg, const Geom& g, Workspace* ws, void* base, size_t avail) {
    const size_t nt = (size_t)n_seg * g.nSlices * g.nTiles;
    const size_t o_table = 0;
    const size_t o_tstart = oess::align_up(o_table + nt * 4, 256);
    const size_t o_total = oess::align_up(o_tstart + (size_t)n_seg * (g.nTiles + 1) * 4, 256);
    const size_t o_base = oess::align_up(o_total + (size_t)n_seg * 4, 256);
    const size_t o_recs = oess::align_up(o_base + (size_t)n_seg * 4, 256);
    const size_t need = o_recs + (size_t)n_events * 4 * sizeof(float4);        // worst case: every event in 4 tiles
    if (ws) {
        char* b = (char*)base;
        ws->table = (int*)(b + o_table); ws->tile_start = (int*)(b + o_tstart); ws->seg_total = (int*)(b + o_total);
        ws->seg_base = (int*)(b + o_base); ws->recs = (float4*)(b + o_recs);
        const size_t rec_bytes = (avail > o_recs) ? avail - o_recs : 0;
        const size_t cap = rec_bytes / sizeof(float4);
        ws->cap = (uint32_t)(cap > 0x7fffffffull ? 0x7fffffffull : cap);
    }
    return need;
}

// tri-linear workspace: [256 B pad] [transposed table: n_seg x (nTiles + 2) x nSl ints] [record regions: n_seg x nSl x RSTRIDE]
Geom tri_geom(int C, int H, int W, int crop_rows, int64_t max_seg_len) {
    // 32-bit accumulators, 20 KB per tile (8 splat workgroups per CU); at least two rows so that the 64-bit path of a dense
    // tile can run as two half-height passes in the same LDS
    Geom g = make_geom(C, H, W, crop_rows, max_seg_len, FAST_LDS_BYTES, 4);      // measured: 40 KB tiles 276 us, 10 KB 350 us, 20 KB 262 us
    if (g.TH < 2) g = make_geom(C, H, W, crop_rows, max_seg_len, 2 * C * TW * 4, 4);
    return g;
}

size_t ws_layout_tri(int64_t n_events, int n_seg, const Geom& g, int nSl, size_t* o_table, size_t* o_recs) {
    const size_t ot = 256;
    const size_t orr = oess::align_up(ot + (size_t)n_seg * nSl * (g.nTiles + 2) * 4, 256);
    if (o_table) *o_table = ot;
    if (o_recs) *o_recs = orr;
    const size_t slots = (size_t)n_seg * nSl * RSTRIDE;                 // fixed record region per (segment, slice)
    const size_t need = (size_t)n_events * 4;
    return orr + (slots > need ? slots : need) * sizeof(float4);
}

template <typename Src>
int run_tri(Src src, const int64_t* seg_off, int n_seg, int64_t max_seg_len, int C, int H, int W, int crop_rows,
            int count_mode, float* out, void* workspace, size_t workspace_bytes, hipStream_t st) {
    using Rec = typename Src::Rec;
    if (n_seg <= 0 || C <= 0 || C > 64 || H <= 0 || W <= 0 || crop_rows < 0 || crop_rows >= H || !out || !seg_off)
        return OESS_EINVAL;
    if (max_seg_len < 0 || max_seg_len > 0x3fffffffll) return OESS_EINVAL;
    Geom g = tri_geom(C, H, W, crop_rows, max_seg_len);
    if (g.nTiles > 8192 || g.TH < 2) return OESS_EINVAL;
    int64_t nsl64 = (max_seg_len + SSL - 1) / SSL;
    if (nsl64 < 1) nsl64 = 1;
    const int nSl = (int)nsl64;
    size_t o_table, o_recs;
    const size_t need = ws_layout_tri(0, n_seg, g, nSl, &o_table, &o_recs);
    if (!workspace || workspace_bytes < need) return OESS_ENOMEM;
    if ((size_t)n_seg * nSl * RSTRIDE > 0xffff0000ull) return OESS_EINVAL;      // 32-bit record indices
    // record capacity of the caller's workspace (oess_voxelize_workspace_bytes sizes it for 4 records per event, the
    // worst case); records beyond it are dropped rather than written out of bounds
    size_t cap_sz = (workspace_bytes - o_recs) / sizeof(Rec);
    if (cap_sz > 0xffff0000ull) cap_sz = 0xffff0000ull;
    const unsigned int cap = (unsigned int)cap_sz;
    char* wb = (char*)workspace;
    int* table = (int*)(wb + o_table);
    Rec* recs = (Rec*)(wb + o_recs);
    Src src_c = src;
    src_c.seg_base_index = 0;
    const size_t sort_lds = (((size_t)(g.nTiles + 1 + 17) * 4 + 15) & ~(size_t)15) + (size_t)LCAP * sizeof(Rec);
    if (sort_lds + 64 > 160 * 1024) return OESS_EINVAL;
    const size_t splat_lds = (size_t)g.C * g.TH * TW * sizeof(int);
    OESS_HIP(hipFuncSetAttribute((const void*)&tri_sort_kernel<Src>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sort_lds));
    hipLaunchKernelGGL((tri_sort_kernel<Src>), dim3(nSl, n_seg), dim3(SORT_THREADS), sort_lds, st, src_c, seg_off, g, nSl, table,
                       recs, cap);
    OESS_HIP(hipFuncSetAttribute((const void*)&tri_splat_kernel<Src>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)splat_lds));
    if (splat_lds < 2048) return OESS_EINVAL;             // the run lookup borrows the first 512 accumulator words
    const long long n_items = (long long)g.nTiles * n_seg;
    if (n_items > 0x7ffffff0ll) return OESS_EINVAL;
    hipLaunchKernelGGL((tri_splat_kernel<Src>), dim3((unsigned)(((n_items + 7) >> 3) << 3)), dim3(THREADS), splat_lds, st, src_c,
                       (const Rec*)recs, (const int*)table, g, nSl, count_mode, cap, (int)n_items, out);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

template <typename T>
int run_near(const T* events, const int64_t* seg_off, int n_seg, int64_t max_seg_len, int nbins, int H, int W,
             int crop_rows, int separate_pol, int count_mode, float* out, void* workspace, size_t workspace_bytes,
             hipStream_t st) {
    if (n_seg <= 0 || nbins <= 0 || nbins > 32 || H <= 0 || W <= 0 || H > 65535 || W > 65535 || crop_rows < 0 ||
        crop_rows >= H || !out || !seg_off || !events)
        return OESS_EINVAL;
    if (max_seg_len < 0 || max_seg_len > 0x3fffffffll) return OESS_EINVAL;
    Geom g = make_geom(2 * nbins, H, W, crop_rows, max_seg_len);
    if (g.nTiles > 8192) return OESS_EINVAL;
    Workspace ws;
    const size_t min_need = ws_layout(0, n_seg, g, &ws, workspace, workspace_bytes);
    if (!workspace || workspace_bytes < min_need) return OESS_ENOMEM;
    SrcNear<T> src{events, nbins};
    const dim3 bin_grid(g.nSlices, n_seg);
    hipLaunchKernelGGL((near_bin_kernel<0, SrcNear<T>>), bin_grid, dim3(THREADS), g.nTiles * sizeof(int), st, src,
                       seg_off, g, ws.table, ws.seg_base, ws.recs, ws.cap);
    hipLaunchKernelGGL(scan_seg_kernel, dim3(n_seg), dim3(1024), 0, st, ws.table, ws.tile_start, ws.seg_total, g);
    hipLaunchKernelGGL(scan_base_kernel, dim3(1), dim3(1024), 0, st, ws.seg_total, ws.seg_base, n_seg);
    hipLaunchKernelGGL((near_bin_kernel<1, SrcNear<T>>), bin_grid, dim3(THREADS), g.nTiles * sizeof(int), st, src,
                       seg_off, g, ws.table, ws.seg_base, ws.recs, ws.cap);
    hipLaunchKernelGGL(near_splat_kernel, dim3(g.nTiles, n_seg), dim3(THREADS),
                       (size_t)g.C * g.TH * TW * sizeof(long long), st, ws.recs, ws.tile_start, ws.seg_base, g, nbins,
                       separate_pol, count_mode, ws.cap, out);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

}  // namespace

extern "C" {

size_t oess_voxelize_workspace_bytes(int64_t n_events, int n_seg, int64_t max_seg_len, int C, int H, int W,
                                     int crop_rows) {
    if (n_events < 0 || n_seg <= 0 || max_seg_len < 0 || C <= 0 || H <= 0 || W <= 0 || crop_rows < 0 || crop_rows >= H)
        return 0;
    // one query serves both pipelines (nearest-xy: 2 x nbins accumulator channels <= C passed by the caller)
    const size_t v1 = ws_layout(n_events, n_seg, make_geom(C, H, W, crop_rows, max_seg_len), nullptr, nullptr, 0);
    int64_t nsl = (max_seg_len + SSL - 1) / SSL;
    if (nsl < 1) nsl = 1;
    const size_t v2 = ws_layout_tri(n_events, n_seg, tri_geom(C, H, W, crop_rows, max_seg_len), (int)nsl, nullptr, nullptr);
    return v1 > v2 ? v1 : v2;
}

int oess_voxelize_trilinear_f32(const float* x, const float* y, const float* p, const float* t,
                                const int64_t* seg_offsets, int n_seg, int64_t max_seg_len, int C, int H, int W,
                                int crop_rows, int count_mode, float* out, void* workspace, size_t workspace_bytes,
                                oess_stream_t stream) {
    if (!x || !y || !p || !t) return OESS_EINVAL;
    SrcF32 src{x, y, p, t, 0};
    return run_tri(src, seg_offsets, n_seg, max_seg_len, C, H, W, crop_rows, count_mode, out, workspace,
                   workspace_bytes, (hipStream_t)stream);
}

int oess_voxelize_dsec_raw(const uint16_t* x, const uint16_t* y, const int64_t* t_us, const uint8_t* p,
                           const float* rectify_maps, const int32_t* seg_map, int n_maps, const int64_t* seg_offsets,
                           int n_seg, int64_t max_seg_len, int C, int H, int W, int crop_rows, int count_mode,
                           float* out, void* workspace, size_t workspace_bytes, oess_stream_t stream) {
    if (!x || !y || !p || !t_us || !rectify_maps || !seg_map || n_maps <= 0) return OESS_EINVAL;
    SrcRaw src{x, y, t_us, p, rectify_maps, seg_map, H, W, 0};
    return run_tri(src, seg_offsets, n_seg, max_seg_len, C, H, W, crop_rows, count_mode, out, workspace,
                   workspace_bytes, (hipStream_t)stream);
}

int oess_voxelize_nearest_i64(const int64_t* events, const int64_t* seg_offsets, int n_seg, int64_t max_seg_len,
                              int nbins, int H, int W, int crop_rows, int separate_pol, int count_mode, float* out,
                              void* workspace, size_t workspace_bytes, oess_stream_t stream) {
    return run_near<long long>((const long long*)events, seg_offsets, n_seg, max_seg_len, nbins, H, W, crop_rows,
                               separate_pol, count_mode, out, workspace, workspace_bytes, (hipStream_t)stream);
}

int oess_voxelize_nearest_f64(const double* events, const int64_t* seg_offsets, int n_seg, int64_t max_seg_len,
                              int nbins, int H, int W, int crop_rows, int separate_pol, int count_mode, float* out,
                              void* workspace, size_t workspace_bytes, oess_stream_t stream) {
    return run_near<double>(events, seg_offsets, n_seg, max_seg_len, nbins, H, W, crop_rows, separate_pol, count_mode,
                            out, workspace, workspace_bytes, (hipStream_t)stream);
}

int oess_event_histogram_i64(const int64_t* events, const int64_t* seg_offsets, int n_seg, int64_t max_seg_len, int H,
                             int W, float* out, oess_stream_t stream) {
    if (!events || !seg_offsets || !out || n_seg <= 0 || H <= 0 || W <= 0 || max_seg_len < 0) return OESS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    OESS_HIP(hipMemsetAsync(out, 0, (size_t)n_seg * 2 * H * W * sizeof(float), st));
    int gx = (int)((max_seg_len + THREADS * 4 - 1) / (THREADS * 4));
    if (gx < 1) gx = 1;
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(hist_kernel, dim3(gx, n_seg), dim3(THREADS), 0, st, events, seg_offsets, H, W, out);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

}  // extern "C"
