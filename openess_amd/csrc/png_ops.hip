// GPU-side decode of the 8-bit single-channel PNG inputs of the hot path: pseudo-label maps
// (`pl_fcclip_rgb/*.png`), superpixel-id maps (`sp_sam_rgb/*.png`, `sp_slic_rgb/*_slic_100.png`) and ground-truth labels, which
// the reference decodes per sample in its loader workers with PIL (`np.array(Image.open(path))`,
// DSEC/dataset/sequence_ov.py:340-358, datasets/ddd17_events_loader.py:228-262) and ships as int64 tensors (8 B / pixel).
// Here the loader hands over the FILE BYTES (a few KB per map); the batch is decoded on the device, on the ingest stream:
//   K1  png_inflate_kernel   one wave per image: chunk walk (IHDR checks, IDAT payloads compacted), zlib header, DEFLATE
//                            (stored / fixed / dynamic blocks; 9-bit direct Huffman tables in LDS, bit-serial canonical tail for
//                            longer codes; the 32 KB LZ77 window is an LDS ring, matches are copied by all 64 lanes) -> the
//                            filtered scanlines in scratch.
//   K2  png_unfilter_kernel  one wave per image: PNG filters None / Sub / Up / Average / Paeth row by row in LDS, then the row leaves
//                            as int64 (the dtype the trainers index with), optionally mirrored (the loader's horizontal-flip
//                            augmentation, sequence_ov.py:366-372), 512 B per wave store.
// Integer / byte work, exact: the result equals PIL's array bit for bit (tests/test_hip_png.py; oracle = PIL + zlib on the host).
// Latency, not throughput, is what this costs (a wave decodes ~20-60 MB/s); the images of a batch decode in parallel on
// separate CUs, on the side stream, under the previous training step.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "oess.h"
#include "oess_common.h"

namespace {

constexpr int WIN = 32768;            // LZ77 window (ring in LDS)
constexpr int FAST_BITS = 9;
constexpr int FAST_N = 1 << FAST_BITS;

enum { ST_OK = 0, ST_BAD_SIGNATURE = 1, ST_BAD_IHDR = 2, ST_UNSUPPORTED = 3, ST_BAD_ZLIB = 4, ST_BAD_BLOCK = 5, ST_BAD_CODE = 6,
       ST_OVERRUN = 7, ST_SIZE_MISMATCH = 8, ST_BAD_FILTER = 9 };

__device__ __forceinline__ uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

// LSB-first bit reader over a contiguous byte range (wave-uniform state: every lane runs the same decode)
struct Bits {
    const uint8_t* p; const uint8_t* end; uint64_t buf; int cnt; int over;
    __device__ __forceinline__ void refill() {
        while (cnt <= 56) {
            uint64_t b = 0;
            if (p < end) b = *p; else ++over;          // past the end: zeros (flagged when more than a tail's worth was needed)
            ++p;
            buf |= b << cnt; cnt += 8;
        }
    }
    __device__ __forceinline__ uint32_t peek(int n) { return (uint32_t)(buf & ((1ull << n) - 1)); }
    __device__ __forceinline__ void drop(int n) { buf >>= n; cnt -= n; }
    __device__ __forceinline__ uint32_t get(int n) { if (cnt < n) refill(); const uint32_t v = peek(n); drop(n); return v; }
};

// canonical Huffman code of `n` symbols with lengths len[] (0 = unused): fast[] = direct table over the next FAST_BITS bits
// (entry = sym << 4 | len, 0 = not a short code), plus count[] / symbol[] for the bit-serial decode of longer codes (puff's form)
struct Huff { uint16_t* fast; uint16_t* count; uint16_t* symbol; };

__device__ void build_huff(const Huff& h, const uint8_t* len, int n, int lane) {
    for (int i = lane; i < FAST_N; i += 64) h.fast[i] = 0;
    if (lane == 0) {
        for (int l = 0; l <= 15; ++l) h.count[l] = 0;
        for (int s = 0; s < n; ++s) h.count[len[s]]++;
        uint16_t offs[16];
        offs[1] = 0;
        for (int l = 1; l < 15; ++l) offs[l + 1] = offs[l] + h.count[l];
        for (int s = 0; s < n; ++s) if (len[s]) h.symbol[offs[len[s]]++] = (uint16_t)s;
        // direct table: canonical codes in symbol order per length, bit-reversed (DEFLATE packs codes MSB-first into an LSB-first stream)
        uint32_t code = 0; int idx = 0;
        for (int l = 1; l <= FAST_BITS; ++l) {
            for (int k = 0; k < h.count[l]; ++k, ++idx, ++code) {
                uint32_t rev = 0;
                for (int b = 0; b < l; ++b) rev |= ((code >> b) & 1u) << (l - 1 - b);
                const uint16_t e = (uint16_t)(h.symbol[idx] << 4 | l);
                for (uint32_t j = rev; j < (uint32_t)FAST_N; j += 1u << l) h.fast[j] = e;
            }
            code <<= 1;
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);       // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int decode_sym(Bits& br, const Huff& h) {
    if (br.cnt < 15) br.refill();
    const uint16_t e = h.fast[br.peek(FAST_BITS)];
    if (e) { br.drop(e & 15); return e >> 4; }
    // longer than FAST_BITS: canonical bit-serial decode (puff.c's loop)
    int code = 0, first = 0, index = 0;
    for (int l = 1; l <= 15; ++l) {
        code |= (int)br.peek(1); br.drop(1);
        const int c = h.count[l];
        if (code - c < first) return h.symbol[index + (code - first)];
        index += c; first += c; first <<= 1; code <<= 1;
    }
    return -1;
}

__constant__ uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__constant__ uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__constant__ uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145,
                                       8193, 12289, 16385, 24577};
__constant__ uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__constant__ uint8_t CL_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// One wave per image.  scratch layout per image: [file bytes: compacted IDAT stream][H * (W + 1): filtered scanlines].
__global__ __launch_bounds__(64) void png_inflate_kernel(const uint8_t* __restrict__ data, const int64_t* __restrict__ offsets, int H, int W,
                                                         uint8_t* __restrict__ scratch, const int64_t* __restrict__ scr_off,
                                                         int* __restrict__ status) {
    __shared__ uint8_t win[WIN];
    __shared__ uint16_t fastL[FAST_N], fastD[FAST_N], cntL[16], cntD[16], symL[288], symD[32];
    __shared__ uint8_t lens[384];            // [0, 19): code-length code; [32, 32 + 286 + 30): literal / distance code lengths
    const int img = blockIdx.x, lane = threadIdx.x;
    const uint8_t* f = data + offsets[img];
    const int64_t flen = offsets[img + 1] - offsets[img];
    uint8_t* zbuf = scratch + scr_off[img];
    uint8_t* raw = zbuf + ((flen + 15) & ~(int64_t)15);
    const int64_t want = (int64_t)H * (W + 1);
    int st = ST_OK;
    // ---- chunk walk
    int64_t zlen = 0;
    {
        const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
        if (flen < 8 + 25) st = ST_BAD_SIGNATURE;
        else for (int i = 0; i < 8; ++i) if (f[i] != sig[i]) st = ST_BAD_SIGNATURE;
        int64_t pos = 8;
        bool seen_ihdr = false;
        while (st == ST_OK && pos + 12 <= flen) {
            const uint32_t clen = be32(f + pos);
            const uint32_t type = be32(f + pos + 4);
            if (pos + 12 + (int64_t)clen > flen) { st = ST_OVERRUN; break; }
            const uint8_t* body = f + pos + 8;
            if (type == 0x49484452u) {                       // IHDR
                if (clen != 13) { st = ST_BAD_IHDR; break; }
                const uint32_t w = be32(body), hh = be32(body + 4);
                if ((int)w != W || (int)hh != H) { st = ST_SIZE_MISMATCH; break; }
                // 8-bit greyscale (0) or 8-bit palette INDICES (3: np.array(Image.open()) of a 'P' image is the index map), no interlace
                if (body[8] != 8 || (body[9] != 0 && body[9] != 3) || body[10] != 0 || body[11] != 0 || body[12] != 0) { st = ST_UNSUPPORTED; break; }
                seen_ihdr = true;
            } else if (type == 0x49444154u) {                // IDAT
                for (int64_t i = lane; i < (int64_t)clen; i += 64) zbuf[zlen + i] = body[i];
                zlen += clen;
            } else if (type == 0x49454e44u) break;           // IEND
            pos += 12 + (int64_t)clen;
        }
        if (st == ST_OK && (!seen_ihdr || zlen < 6)) st = ST_BAD_IHDR;
    }
    __threadfence_block();
    __builtin_amdgcn_s_waitcnt(0);          // the compacted stream is re-read below (same wave, same CU: write-through L1)
    __builtin_amdgcn_wave_barrier();
    int64_t out_n = 0;
    if (st == ST_OK) {
        Bits br{zbuf + 2, zbuf + zlen, 0, 0, 0};
        if ((zbuf[0] & 15) != 8 || (((uint32_t)zbuf[0] << 8 | zbuf[1]) % 31u) != 0 || (zbuf[1] & 32)) st = ST_BAD_ZLIB;
        const Huff HL{fastL, cntL, symL}, HD{fastD, cntD, symD};
        auto emit = [&](uint8_t b) {           // uniform: lane 0 stores
            if (out_n < want && lane == 0) { win[out_n & (WIN - 1)] = b; raw[out_n] = b; }
            ++out_n;
        };
        bool last = false;
        while (st == ST_OK && !last) {
            if (br.over > 8) { st = ST_OVERRUN; break; }
            last = br.get(1) != 0;
            const uint32_t bt = br.get(2);
            if (bt == 0) {                                  // stored
                br.drop(br.cnt & 7);
                const uint32_t ln = br.get(16), nl = br.get(16);
                if ((ln ^ 0xffffu) != nl) { st = ST_BAD_BLOCK; break; }
                // a stored block longer than what is left of the stream or of the image is corrupt: stop before copying it
                if (out_n + (int64_t)ln > want) { st = ST_SIZE_MISMATCH; break; }
                if ((int64_t)ln > (br.end - br.p) + (br.cnt >> 3)) { st = ST_OVERRUN; break; }
                for (uint32_t i = 0; i < ln; ++i) emit((uint8_t)br.get(8));
                continue;
            }
            if (bt == 3) { st = ST_BAD_BLOCK; break; }
            int nl = 288, nd = 30;
            if (bt == 1) {
                for (int i = lane; i < 288; i += 64) lens[i] = i < 144 ? 8 : (i < 256 ? 9 : (i < 280 ? 7 : 8));
                __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier();
                build_huff(HL, lens, 288, lane);
                for (int i = lane; i < 30; i += 64) lens[i] = 5;
                __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier();
                build_huff(HD, lens, 30, lane);
            } else {
                nl = (int)br.get(5) + 257; nd = (int)br.get(5) + 1;
                const int ncl = (int)br.get(4) + 4;
                if (nl > 286 || nd > 30) { st = ST_BAD_BLOCK; break; }
                for (int i = lane; i < 19; i += 64) lens[i] = 0;
                __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier();
                for (int i = 0; i < ncl; ++i) { const uint8_t v = (uint8_t)br.get(3); if (lane == 0) lens[CL_ORDER[i]] = v; }
                __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier();
                build_huff(HL, lens, 19, lane);            // code-length code, in the literal table's storage
                int i = 0;
                // code lengths of both alphabets, run-length coded (written by lane 0, read back uniformly through LDS)
                while (i < nl + nd && st == ST_OK) {
                    if (br.over > 8) { st = ST_OVERRUN; break; }
                    const int sym = decode_sym(br, HL);
                    if (sym < 0) { st = ST_BAD_CODE; break; }
                    if (sym < 16) { if (lane == 0) lens[32 + i] = (uint8_t)sym; ++i; }
                    else {
                        int rep, val = 0;
                        if (sym == 16) {
                            if (i == 0) { st = ST_BAD_CODE; break; }
                            __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier();
                            val = lens[32 + i - 1]; rep = 3 + (int)br.get(2);
                        } else if (sym == 17) rep = 3 + (int)br.get(3);
                        else rep = 11 + (int)br.get(7);
                        if (i + rep > nl + nd) { st = ST_BAD_CODE; break; }
                        if (lane == 0) for (int k = 0; k < rep; ++k) lens[32 + i + k] = (uint8_t)val;
                        i += rep;
                    }
                }
                if (st != ST_OK) break;
                __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier();
                build_huff(HL, lens + 32, nl, lane);
                build_huff(HD, lens + 32 + nl, nd, lane);
            }
            // ---- symbols of the block
            // Every iteration is bounded: once the reader has run more than a tail's worth past the stream (it feeds zeros there,
            // and the all-zero code is usually a literal, so the loop would never end on its own) or the image is over-full,
            // the file is corrupt.  Chunk CRC / Adler-32 are not verified, so this is what stops a truncated zlib stream.
            while (st == ST_OK) {
                if (br.over > 8) { st = ST_OVERRUN; break; }
                if (out_n > want) { st = ST_SIZE_MISMATCH; break; }
                int sym = decode_sym(br, HL);
                if (sym < 0) { st = ST_BAD_CODE; break; }
                if (sym < 256) { emit((uint8_t)sym); continue; }
                if (sym == 256) break;
                sym -= 257;
                if (sym >= 29) { st = ST_BAD_CODE; break; }
                const int mlen = LEN_BASE[sym] + (int)br.get(LEN_EXTRA[sym]);
                const int ds = decode_sym(br, HD);
                if (ds < 0 || ds >= 30) { st = ST_BAD_CODE; break; }
                const int dist = DIST_BASE[ds] + (int)br.get(DIST_EXTRA[ds]);
                if (dist > out_n || dist > WIN) { st = ST_BAD_CODE; break; }
                // match copy by all lanes; an overlapping match (dist < mlen) is the periodic extension of the last `dist` bytes
                __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier();
                for (int i0 = 0; i0 < mlen; i0 += 64) {
                    const int i = i0 + lane;
                    if (i < mlen && out_n + i < want) {
                        const uint8_t b = win[(out_n - dist + (i % dist)) & (WIN - 1)];
                        raw[out_n + i] = b;
                        // the window slot of byte out_n + i can only be a source of THIS match if dist > WIN - mlen; such sources
                        // were read above (i % dist indexes the bytes before out_n), so writing after the read is safe per iteration
                        win[(out_n + i) & (WIN - 1)] = b;
                    }
                    __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier();
                }
                out_n += mlen;
            }
        }
        if (st == ST_OK && br.over > 8) st = ST_OVERRUN;
        if (st == ST_OK && out_n != want) st = ST_SIZE_MISMATCH;
    }
    if (lane == 0) status[img] = st;
}

__device__ __forceinline__ int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// One wave per image: unfilter row by row (two row buffers in LDS), then the row leaves as int64, optionally mirrored.
__global__ __launch_bounds__(64) void png_unfilter_kernel(const uint8_t* __restrict__ scratch, const int64_t* __restrict__ scr_off,
                                                          const int64_t* __restrict__ offsets, int H, int W,
                                                          const uint8_t* __restrict__ flip, int64_t* __restrict__ out,
                                                          int* __restrict__ status) {
    extern __shared__ uint8_t rows[];                    // [2][W]
    const int img = blockIdx.x, lane = threadIdx.x;
    const int64_t flen = offsets[img + 1] - offsets[img];
    const uint8_t* raw = scratch + scr_off[img] + ((flen + 15) & ~(int64_t)15);
    int64_t* o = out + (int64_t)img * H * W;
    if (status[img] != ST_OK) {                           // undecodable file: the map is all 255 (ignore index), the status says why
        for (int64_t i = lane; i < (int64_t)H * W; i += 64) o[i] = 255;
        return;
    }
    const bool mirror = flip && flip[img];
    uint8_t* prev = rows; uint8_t* cur = rows + W;
    for (int x = lane; x < W; x += 64) prev[x] = 0;
    int bad = 0;
    for (int y = 0; y < H; ++y) {
        const uint8_t* fr = raw + (int64_t)y * (W + 1);
        const int ft = fr[0];
        __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier();
        if (ft == 0) { for (int x = lane; x < W; x += 64) cur[x] = fr[1 + x]; }
        else if (ft == 2) { for (int x = lane; x < W; x += 64) cur[x] = (uint8_t)(fr[1 + x] + prev[x]); }
        else if (ft == 1 || ft == 3 || ft == 4) {
            // serial along x (Average and Paeth are not associative); the chain lives in registers, the filtered bytes and the
            // previous row are independent loads the compiler batches ahead of it.  Every lane runs the same loop; lane 0 writes.
            int left = 0, upleft = 0;
            for (int x = 0; x < W; ++x) {
                const int up = prev[x];
                int v = fr[1 + x];
                if (ft == 1) v += left; else if (ft == 3) v += (left + up) >> 1; else v += paeth(left, up, upleft);
                v &= 255;
                if (lane == 0) cur[x] = (uint8_t)v;
                left = v; upleft = up;
            }
        } else { bad = 1; for (int x = lane; x < W; x += 64) cur[x] = 255; }
        __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier();
        int64_t* orow = o + (int64_t)y * W;
        for (int x = lane; x < W; x += 64) orow[x] = (int64_t)cur[mirror ? W - 1 - x : x];
        uint8_t* t = prev; prev = cur; cur = t;
    }
    if (bad && lane == 0) status[img] = ST_BAD_FILTER;
}

}  // namespace

extern "C" {

size_t oess_png_decode_scratch_bytes(long long total_file_bytes, int n_images, int H, int W) {
    if (total_file_bytes < 0 || n_images <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)total_file_bytes + (size_t)n_images * (16 + (size_t)H * (W + 1) + 16) + (size_t)(n_images + 1) * 8 + 256;
}

int oess_png_decode_gray8_batch(const uint8_t* files, const int64_t* offsets, long long total_file_bytes, int n_images, int H, int W,
                                const uint8_t* flip, int64_t* out, void* scratch, size_t scratch_bytes, const int64_t* scratch_offsets,
                                int* status, oess_stream_t stream) {
    if (!files || !offsets || !out || !scratch || !scratch_offsets || !status || n_images <= 0 || H <= 0 || W <= 0 || W > 16384 ||
        total_file_bytes <= 0)
        return OESS_EINVAL;
    if (scratch_bytes < oess_png_decode_scratch_bytes(total_file_bytes, n_images, H, W)) return OESS_ENOMEM;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(png_inflate_kernel, dim3(n_images), dim3(64), 0, st, files, offsets, H, W, (uint8_t*)scratch, scratch_offsets, status);
    hipLaunchKernelGGL(png_unfilter_kernel, dim3(n_images), dim3(64), (size_t)2 * W, st, (const uint8_t*)scratch, scratch_offsets, offsets, H, W,
                       flip, out, status);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

}  // extern "C"
