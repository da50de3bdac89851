// Event -> voxel-grid builders for gfx950 (K1 tri-linear, K1' nearest-xy, event histogram).
//
// Design (MI355X-first, HBM-bound integer/scatter work -- no MFMA):
//   The reference scatters 8 (tri-linear) or 2 (nearest) read-modify-writes per event into a
//   C x H x W grid.  Global fp32 atomics across the 8 non-coherent XCD L2s would have to execute
//   memory-side; instead the splat is made OUTPUT-STATIONARY:
//     A  count    events per (segment, spatial tile)            LDS histogram -> global counters
//     B  scan     exclusive prefix over (segment, tile)         one workgroup
//     C  scatter  events -> tile-binned 16-byte records          LDS rank + one global atomic per
//                                                                (workgroup, tile)
//     D  splat    one workgroup per (segment, tile): records -> LDS fp32 atomics (ds_add_f32)
//                 into a C x TH x 64 tile, then every output voxel is written ONCE, coalesced
//                 (float4 per lane).  No pre-zeroing pass, no global atomics on the grid.
//   An event whose 2x2 pixel footprint straddles a tile edge is binned into each tile it touches
//   (<= 4, ~5 % duplication at 64x32 tiles); each tile only accumulates the corners it owns.
//
// Parity: per-event index math and weights follow the reference's float32 / float64 operation
// order exactly (compiled with -ffp-contract=off; IEEE division), so indices are bit-exact and
// every individual contribution is bit-identical; only the summation ORDER differs (LDS atomics).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "oess.h"
#include "oess_common.h"

namespace {

constexpr int TW = 64;            // tile width (pixels) = one wave of lanes
constexpr int THREADS = 256;
constexpr int EPT = 8;            // events per thread in count / scatter
constexpr int MAX_LDS_TILE_BYTES = 40 * 1024;   // 4 workgroups per CU (160 KiB LDS)

struct Geom {
    int C;        // channels accumulated in LDS per tile (tri-linear: bins; nearest: 2*bins)
    int H, W;     // sensor size used for the reference's validity masks
    int Hout;     // rows kept (H - crop_rows)
    int TH;       // tile height
    int tilesX, tilesY, nTiles;
};

__host__ Geom make_geom(int C, int H, int W, int crop_rows) {
    Geom g;
    g.C = C; g.H = H; g.W = W; g.Hout = H - crop_rows;
    int th = MAX_LDS_TILE_BYTES / (C * TW * 4);
    if (th > 32) th = 32;
    if (th < 1) th = 1;
    g.TH = th;
    g.tilesX = (W + TW - 1) / TW;
    g.tilesY = (g.Hout + th - 1) / th;
    g.nTiles = g.tilesX * g.tilesY;
    return g;
}

// ---------------------------------------------------------------------------------------------
// Event sources.  load() returns false when the event contributes nothing at all.
// rec = {x, y, t_norm, value} for the tri-linear splat.
// ---------------------------------------------------------------------------------------------
struct TriRec { float x, y, tn, v; };

struct SrcF32 {                         // VoxelGrid.convert's own arguments
    const float* x; const float* y; const float* p; const float* t;
    struct Seg { float t0, denom; };
    __device__ Seg seg(int /*s*/, int64_t b, int64_t e) const {
        Seg sg; sg.t0 = t[b]; sg.denom = __fsub_rn(t[e - 1], sg.t0); return sg;
    }
    __device__ TriRec load(int64_t i, const Seg& sg, int C) const {
        TriRec r;
        r.x = x[i]; r.y = y[i];
        // representations.py:25  (C-1)*(t-t[0]) / (t[-1]-t[0])   float32, left to right
        r.tn = __fmul_rn((float)(C - 1), __fsub_rn(t[i], sg.t0)) / sg.denom;
        r.v = __fsub_rn(__fmul_rn(2.0f, p[i]), 1.0f);             // representations.py:31
        return r;
    }
};

struct SrcRaw {                         // raw DSEC columns + rectify map (sequence_ov.py:154-157,204-210)
    const uint16_t* x; const uint16_t* y; const int64_t* t; const uint8_t* p;
    const float* maps; const int32_t* seg_map; int H, W;
    struct Seg { int64_t t0; float dlast; float tn0, denom; const float* map; };
    __device__ Seg seg(int s, int64_t b, int64_t e) const {
        Seg sg; sg.t0 = t[b];
        sg.dlast = (float)(double)(t[e - 1] - sg.t0);              // (t-t[0]).astype('float32')[-1]
        float first = 0.0f / sg.dlast;                            // t/t[-1] at index 0 (NaN if dlast==0)
        float last = sg.dlast / sg.dlast;
        sg.tn0 = first; sg.denom = __fsub_rn(last, first);
        sg.map = maps + (size_t)seg_map[s] * (size_t)H * W * 2;
        return sg;
    }
    __device__ TriRec load(int64_t i, const Seg& sg, int C) const {
        TriRec r;
        int xi = x[i], yi = y[i];
        const float2 m = *reinterpret_cast<const float2*>(sg.map + ((size_t)yi * W + xi) * 2);
        r.x = m.x; r.y = m.y;
        float tt = (float)(double)(t[i] - sg.t0) / sg.dlast;
        r.tn = __fmul_rn((float)(C - 1), __fsub_rn(tt, sg.tn0)) / sg.denom;
        r.v = __fsub_rn(__fmul_rn(2.0f, (float)p[i]), 1.0f);
        return r;
    }
};

// Tiles touched by a tri-linear event.  Returns the number of tiles (0..4) in tiles[].
__device__ __forceinline__ int tri_tiles(const TriRec& r, const Geom& g, int tiles[4]) {
    if (!(fabsf(r.tn) < 1.0e9f)) return 0;          // NaN/inf time: Tensor.int() gives INT_MIN on the CPU -> all masked
    int t0 = (int)r.tn;
    if (!((t0 >= 0 && t0 < g.C) || (t0 + 1 >= 0 && t0 + 1 < g.C))) return 0;
    if (r.x != r.x || r.y != r.y) return 0;
    // clamp before the int conversion so that huge coordinates stay "far outside" instead of UB
    float fx = fminf(fmaxf(r.x, -8.0f), (float)g.W + 8.0f);
    float fy = fminf(fmaxf(r.y, -8.0f), (float)g.H + 8.0f);
    int x0 = (int)fx, y0 = (int)fy;                 // C-style truncation (representations.py:27-28)
    int cx[2], cy[2], ncx = 0, ncy = 0;
    if (x0 >= 0 && x0 < g.W) cx[ncx++] = x0 / TW;
    if (x0 + 1 >= 0 && x0 + 1 < g.W) { int c = (x0 + 1) / TW; if (ncx == 0 || cx[0] != c) cx[ncx++] = c; }
    if (y0 >= 0 && y0 < g.Hout) cy[ncy++] = y0 / g.TH;
    if (y0 + 1 >= 0 && y0 + 1 < g.Hout) { int c = (y0 + 1) / g.TH; if (ncy == 0 || cy[0] != c) cy[ncy++] = c; }
    int n = 0;
    for (int a = 0; a < ncy; ++a)
        for (int b = 0; b < ncx; ++b) tiles[n++] = cy[a] * g.tilesX + cx[b];
    return n;
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
        tile = (int)(ys / g.TH) * g.tilesX + (int)(xs / TW);
        return true;
    }
};

// ---------------------------------------------------------------------------------------------
// Pass A: count  /  Pass C: scatter   (one template, MODE 0 = count, 1 = scatter)
// ---------------------------------------------------------------------------------------------
template <int MODE, typename Src>
__global__ __launch_bounds__(THREADS) void tri_bin_kernel(Src src, const int64_t* __restrict__ seg_off, Geom g,
                                                          int* __restrict__ counts, int* __restrict__ cursor,
                                                          float4* __restrict__ recs, uint32_t cap) {
    extern __shared__ int lds[];          // [nTiles] histogram (+ [nTiles] base in scatter mode)
    const int s = blockIdx.y;
    const int64_t b = seg_off[s], e = seg_off[s + 1];
    const int64_t n = e - b;
    const int64_t first = (int64_t)blockIdx.x * (THREADS * EPT);
    if (first >= n) return;
    for (int i = threadIdx.x; i < g.nTiles; i += THREADS) lds[i] = 0;
    __syncthreads();
    const typename Src::Seg sg = src.seg(s, b, e);
    TriRec rec[EPT];
    uint32_t slot[EPT][4];
    int nt[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int64_t i = first + k * THREADS + threadIdx.x;
        nt[k] = 0;
        if (i < n) {
            rec[k] = src.load(b + i, sg, g.C);
            int tiles[4];
            nt[k] = tri_tiles(rec[k], g, tiles);
            for (int j = 0; j < nt[k]; ++j) {
                int rank = atomicAdd(&lds[tiles[j]], 1);
                slot[k][j] = ((uint32_t)tiles[j] << 16) | (uint32_t)rank;
            }
        }
    }
    __syncthreads();
    if (MODE == 0) {
        for (int i = threadIdx.x; i < g.nTiles; i += THREADS)
            if (lds[i]) atomicAdd(&counts[(size_t)s * g.nTiles + i], lds[i]);
    } else {
        int* base = lds + g.nTiles;
        for (int i = threadIdx.x; i < g.nTiles; i += THREADS)
            base[i] = lds[i] ? atomicAdd(&cursor[(size_t)s * g.nTiles + i], lds[i]) : 0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < EPT; ++k)
            for (int j = 0; j < nt[k]; ++j) {
                uint32_t pos = (uint32_t)base[slot[k][j] >> 16] + (slot[k][j] & 0xffffu);
                if (pos < cap) recs[pos] = make_float4(rec[k].x, rec[k].y, rec[k].tn, rec[k].v);
            }
    }
}

template <int MODE, typename Src>
__global__ __launch_bounds__(THREADS) void near_bin_kernel(Src src, const int64_t* __restrict__ seg_off, Geom g,
                                                           int* __restrict__ counts, int* __restrict__ cursor,
                                                           float4* __restrict__ recs, uint32_t cap) {
    extern __shared__ int lds[];
    const int s = blockIdx.y;
    const int64_t b = seg_off[s], e = seg_off[s + 1];
    const int64_t n = e - b;
    const int64_t first = (int64_t)blockIdx.x * (THREADS * EPT);
    if (first >= n) return;
    for (int i = threadIdx.x; i < g.nTiles; i += THREADS) lds[i] = 0;
    __syncthreads();
    const typename Src::Seg sg = src.seg(s, b, e);
    NearRec rec[EPT];
    uint32_t slot[EPT];
    bool ok[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int64_t i = first + k * THREADS + threadIdx.x;
        ok[k] = false;
        if (i < n) {
            int tile;
            ok[k] = src.load(b + i, sg, g, rec[k], tile);
            if (ok[k]) {
                int rank = atomicAdd(&lds[tile], 1);
                slot[k] = ((uint32_t)tile << 16) | (uint32_t)rank;
            }
        }
    }
    __syncthreads();
    if (MODE == 0) {
        for (int i = threadIdx.x; i < g.nTiles; i += THREADS)
            if (lds[i]) atomicAdd(&counts[(size_t)s * g.nTiles + i], lds[i]);
    } else {
        int* base = lds + g.nTiles;
        for (int i = threadIdx.x; i < g.nTiles; i += THREADS)
            base[i] = lds[i] ? atomicAdd(&cursor[(size_t)s * g.nTiles + i], lds[i]) : 0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < EPT; ++k)
            if (ok[k]) {
                uint32_t pos = (uint32_t)base[slot[k] >> 16] + (slot[k] & 0xffffu);
                if (pos < cap)
                    recs[pos] = make_float4(__uint_as_float(rec[k].xy), __uint_as_float(rec[k].tp), rec[k].vl, rec[k].vr);
            }
    }
}

// ---------------------------------------------------------------------------------------------
// Pass B: exclusive scan of counts -> offsets (and a copy into cursor).  One workgroup.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void scan_kernel(const int* __restrict__ counts, int* __restrict__ offsets,
                                                    int* __restrict__ cursor, int n) {
    __shared__ int part[1024];
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        int i = base + threadIdx.x;
        int v = (i < n) ? counts[i] : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {          // Hillis-Steele inclusive scan
            int a = (threadIdx.x >= off) ? part[threadIdx.x - off] : 0;
            __syncthreads();
            part[threadIdx.x] += a;
            __syncthreads();
        }
        int excl = part[threadIdx.x] - v + carry_s;
        if (i < n) { offsets[i] = excl; cursor[i] = excl; }
        __syncthreads();
        if (threadIdx.x == 1023) carry_s += part[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) offsets[n] = carry_s;
}

// ---------------------------------------------------------------------------------------------
// Pass D: splat one (segment, tile) into LDS, then write every voxel of the tile once.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void write_tile(const float* acc, float* __restrict__ out, const Geom& g, int s,
                                           int ch_out, int tx, int ty, bool diff_pol) {
    // acc layout [C][TH][TW]; out layout [(s*ch_out + c)][Hout][W]
    const int x_base = tx * TW, y_base = ty * g.TH;
    const size_t plane = (size_t)g.Hout * g.W;
    const int nb = diff_pol ? g.C / 2 : 0;
    if ((g.W & 3) == 0) {
        const int q_per_row = TW / 4;
        const int total = ch_out * g.TH * q_per_row;
        for (int i = threadIdx.x; i < total; i += THREADS) {
            int q = i % q_per_row, rr = (i / q_per_row) % g.TH, c = i / (q_per_row * g.TH);
            int xx = x_base + q * 4, yy = y_base + rr;
            if (xx < g.W && yy < g.Hout) {
                float4 v = *reinterpret_cast<const float4*>(&acc[(c * g.TH + rr) * TW + q * 4]);
                if (diff_pol) {
                    float4 m = *reinterpret_cast<const float4*>(&acc[((c + nb) * g.TH + rr) * TW + q * 4]);
                    v.x -= m.x; v.y -= m.y; v.z -= m.z; v.w -= m.w;     // voxel_grid_positive - negative
                }
                *reinterpret_cast<float4*>(&out[((size_t)s * ch_out + c) * plane + (size_t)yy * g.W + xx]) = v;
            }
        }
    } else {
        const int total = ch_out * g.TH * TW;
        for (int i = threadIdx.x; i < total; i += THREADS) {
            int q = i % TW, rr = (i / TW) % g.TH, c = i / (TW * g.TH);
            int xx = x_base + q, yy = y_base + rr;
            if (xx < g.W && yy < g.Hout) {
                float v = acc[(c * g.TH + rr) * TW + q];
                if (diff_pol) v -= acc[((c + nb) * g.TH + rr) * TW + q];
                out[((size_t)s * ch_out + c) * plane + (size_t)yy * g.W + xx] = v;
            }
        }
    }
}

__global__ __launch_bounds__(THREADS) void tri_splat_kernel(const float4* __restrict__ recs,
                                                            const int* __restrict__ offsets, Geom g, int count_mode,
                                                            uint32_t cap, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float acc[];
    const int tile = blockIdx.x, s = blockIdx.y;
    const int tx = tile % g.tilesX, ty = tile / g.tilesX;
    const int lds_n = g.C * g.TH * TW;
    for (int i = threadIdx.x; i < lds_n; i += THREADS) acc[i] = 0.0f;
    __syncthreads();
    const uint32_t beg = (uint32_t)offsets[(size_t)s * g.nTiles + tile];
    uint32_t end = (uint32_t)offsets[(size_t)s * g.nTiles + tile + 1];
    if (end > cap) end = cap;
    const int x_lo = tx * TW, y_lo = ty * g.TH;
    for (uint32_t i = beg + threadIdx.x; i < end; i += THREADS) {
        const float4 r = recs[i];
        const float x = r.x, y = r.y, tn = r.z, val = r.w;
        float fx = fminf(fmaxf(x, -8.0f), (float)g.W + 8.0f);
        float fy = fminf(fmaxf(y, -8.0f), (float)g.H + 8.0f);
        const int x0 = (int)fx, y0 = (int)fy, t0 = (int)tn;
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int xl = x0 + dx;
            const int lx = xl - x_lo;
            if (xl < 0 || xl >= g.W || lx < 0 || lx >= TW) continue;
            const float wx = __fmul_rn(val, __fsub_rn(1.0f, fabsf(__fsub_rn((float)xl, x))));
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
                const int yl = y0 + dy;
                const int ly = yl - y_lo;
                if (yl < 0 || yl >= g.Hout || ly < 0 || ly >= g.TH) continue;
                const float wxy = __fmul_rn(wx, __fsub_rn(1.0f, fabsf(__fsub_rn((float)yl, y))));
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const int tl = t0 + dt;
                    if (tl < 0 || tl >= g.C) continue;
                    float w = __fmul_rn(wxy, __fsub_rn(1.0f, fabsf(__fsub_rn((float)tl, tn))));
                    if (count_mode) w = 1.0f;
                    atomicAdd(&acc[(tl * g.TH + ly) * TW + lx], w);
                }
            }
        }
    }
    __syncthreads();
    write_tile(acc, out, g, s, g.C, tx, ty, false);
}

__global__ __launch_bounds__(THREADS) void near_splat_kernel(const float4* __restrict__ recs,
                                                             const int* __restrict__ offsets, Geom g, int nbins,
                                                             int separate_pol, int count_mode, uint32_t cap,
                                                             float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float acc[];   // [2*nbins][TH][TW]: pos bins then neg bins
    const int tile = blockIdx.x, s = blockIdx.y;
    const int tx = tile % g.tilesX, ty = tile / g.tilesX;
    const int lds_n = g.C * g.TH * TW;
    for (int i = threadIdx.x; i < lds_n; i += THREADS) acc[i] = 0.0f;
    __syncthreads();
    const uint32_t beg = (uint32_t)offsets[(size_t)s * g.nTiles + tile];
    uint32_t end = (uint32_t)offsets[(size_t)s * g.nTiles + tile + 1];
    if (end > cap) end = cap;
    const int x_lo = tx * TW, y_lo = ty * g.TH;
    for (uint32_t i = beg + threadIdx.x; i < end; i += THREADS) {
        const float4 r = recs[i];
        const uint32_t xy = __float_as_uint(r.x), tp = __float_as_uint(r.y);
        const int lx = (int)(xy & 0xffffu) - x_lo, ly = (int)(xy >> 16) - y_lo;
        const int tis = (int)(tp & 0x7fffffffu);
        const int pol_base = (tp & 0x80000000u) ? 0 : nbins;
        float vl = r.z, vr = r.w;
        if (count_mode) { vl = 1.0f; vr = 1.0f; }
        if (tis < nbins) atomicAdd(&acc[((pol_base + tis) * g.TH + ly) * TW + lx], vl);          // data_util.py:86-93
        if (tis + 1 < nbins) atomicAdd(&acc[((pol_base + tis + 1) * g.TH + ly) * TW + lx], vr);  // data_util.py:95-98
    }
    __syncthreads();
    write_tile(acc, out, g, s, separate_pol ? 2 * nbins : nbins, tx, ty, !separate_pol);
}

// Event histogram: one workgroup per (segment, tile); tiny (a4), direct scan of the segment.
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
    int* counts; int* offsets; int* cursor; float4* recs; uint32_t cap;
};

size_t ws_layout(int64_t n_events, int n_seg, const Geom& g, Workspace* ws, void* base, size_t avail) {
    size_t nt = (size_t)n_seg * g.nTiles;
    size_t o_counts = 0;
    size_t o_offsets = oess::align_up(o_counts + nt * 4, 256);
    size_t o_cursor = oess::align_up(o_offsets + (nt + 1) * 4, 256);
    size_t o_recs = oess::align_up(o_cursor + nt * 4, 256);
    size_t need = o_recs + (size_t)n_events * 4 * sizeof(float4);        // worst case: every event in 4 tiles
    if (ws) {
        char* b = (char*)base;
        ws->counts = (int*)(b + o_counts); ws->offsets = (int*)(b + o_offsets); ws->cursor = (int*)(b + o_cursor);
        ws->recs = (float4*)(b + o_recs);
        size_t rec_bytes = (avail > o_recs) ? avail - o_recs : 0;
        size_t cap = rec_bytes / sizeof(float4);
        ws->cap = (uint32_t)(cap > 0x7fffffffull ? 0x7fffffffull : cap);
    }
    return need;
}

template <typename Src>
int run_tri(Src src, const int64_t* seg_off, int n_seg, int64_t max_seg_len, int64_t n_events_hint, int C, int H, int W,
            int crop_rows, int count_mode, float* out, void* workspace, size_t workspace_bytes, hipStream_t st) {
    if (n_seg <= 0 || C <= 0 || C > 64 || H <= 0 || W <= 0 || crop_rows < 0 || crop_rows >= H || !out || !seg_off)
        return OESS_EINVAL;
    if (max_seg_len < 0 || max_seg_len > 0x3fffffffll) return OESS_EINVAL;
    Geom g = make_geom(C, H, W, crop_rows);
    if (g.nTiles > 65535) return OESS_EINVAL;
    Workspace ws;
    size_t min_need = ws_layout(0, n_seg, g, &ws, workspace, workspace_bytes);
    if (!workspace || workspace_bytes < min_need) return OESS_ENOMEM;
    (void)n_events_hint;
    const size_t nt = (size_t)n_seg * g.nTiles;
    OESS_HIP(hipMemsetAsync(ws.counts, 0, nt * sizeof(int), st));
    const int gx = (int)((max_seg_len + THREADS * EPT - 1) / (THREADS * EPT));
    if (gx > 0) {
        dim3 grid(gx, n_seg);
        hipLaunchKernelGGL((tri_bin_kernel<0, Src>), grid, dim3(THREADS), g.nTiles * sizeof(int), st, src, seg_off, g,
                           ws.counts, ws.cursor, ws.recs, ws.cap);
    }
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, st, ws.counts, ws.offsets, ws.cursor, (int)nt);
    if (gx > 0) {
        dim3 grid(gx, n_seg);
        hipLaunchKernelGGL((tri_bin_kernel<1, Src>), grid, dim3(THREADS), 2 * g.nTiles * sizeof(int), st, src, seg_off,
                           g, ws.counts, ws.cursor, ws.recs, ws.cap);
    }
    hipLaunchKernelGGL(tri_splat_kernel, dim3(g.nTiles, n_seg), dim3(THREADS), (size_t)g.C * g.TH * TW * sizeof(float), st,
                       ws.recs, ws.offsets, g, count_mode, ws.cap, out);
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
    Geom g = make_geom(2 * nbins, H, W, crop_rows);
    if (g.nTiles > 65535) return OESS_EINVAL;
    Workspace ws;
    size_t min_need = ws_layout(0, n_seg, g, &ws, workspace, workspace_bytes);
    if (!workspace || workspace_bytes < min_need) return OESS_ENOMEM;
    SrcNear<T> src{events, nbins};
    const size_t nt = (size_t)n_seg * g.nTiles;
    OESS_HIP(hipMemsetAsync(ws.counts, 0, nt * sizeof(int), st));
    const int gx = (int)((max_seg_len + THREADS * EPT - 1) / (THREADS * EPT));
    if (gx > 0) {
        dim3 grid(gx, n_seg);
        hipLaunchKernelGGL((near_bin_kernel<0, SrcNear<T>>), grid, dim3(THREADS), g.nTiles * sizeof(int), st, src, seg_off,
                           g, ws.counts, ws.cursor, ws.recs, ws.cap);
    }
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, st, ws.counts, ws.offsets, ws.cursor, (int)nt);
    if (gx > 0) {
        dim3 grid(gx, n_seg);
        hipLaunchKernelGGL((near_bin_kernel<1, SrcNear<T>>), grid, dim3(THREADS), 2 * g.nTiles * sizeof(int), st, src,
                           seg_off, g, ws.counts, ws.cursor, ws.recs, ws.cap);
    }
    hipLaunchKernelGGL(near_splat_kernel, dim3(g.nTiles, n_seg), dim3(THREADS), (size_t)g.C * g.TH * TW * sizeof(float),
                       st, ws.recs, ws.offsets, g, nbins, separate_pol, count_mode, ws.cap, out);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

}  // namespace

extern "C" {

size_t oess_voxelize_workspace_bytes(int64_t n_events, int n_seg, int C, int H, int W, int crop_rows) {
    if (n_events < 0 || n_seg <= 0 || C <= 0 || H <= 0 || W <= 0 || crop_rows < 0 || crop_rows >= H) return 0;
    Geom g = make_geom(C, H, W, crop_rows);
    return ws_layout(n_events, n_seg, g, nullptr, nullptr, 0);
}

int oess_voxelize_trilinear_f32(const float* x, const float* y, const float* p, const float* t,
                                const int64_t* seg_offsets, int n_seg, int64_t max_seg_len, int C, int H, int W,
                                int crop_rows, int count_mode, float* out, void* workspace, size_t workspace_bytes,
                                oess_stream_t stream) {
    if (!x || !y || !p || !t) return OESS_EINVAL;
    SrcF32 src{x, y, p, t};
    return run_tri(src, seg_offsets, n_seg, max_seg_len, 0, C, H, W, crop_rows, count_mode, out, workspace,
                   workspace_bytes, (hipStream_t)stream);
}

int oess_voxelize_dsec_raw(const uint16_t* x, const uint16_t* y, const int64_t* t_us, const uint8_t* p,
                           const float* rectify_maps, const int32_t* seg_map, int n_maps, const int64_t* seg_offsets,
                           int n_seg, int64_t max_seg_len, int C, int H, int W, int crop_rows, int count_mode,
                           float* out, void* workspace, size_t workspace_bytes, oess_stream_t stream) {
    if (!x || !y || !p || !t_us || !rectify_maps || !seg_map || n_maps <= 0) return OESS_EINVAL;
    SrcRaw src{x, y, t_us, p, rectify_maps, seg_map, H, W};
    return run_tri(src, seg_offsets, n_seg, max_seg_len, 0, C, H, W, crop_rows, count_mode, out, workspace,
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
