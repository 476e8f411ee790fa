// quant.hip — quantized optimizer state: int16 statistics / preconditioners with
// the diagonal kept in float32, int8 momentum.  Follows QuantizedValue.quantize / to_float of the
// reference (precondition/quantization_utils.py:45-95 and :97-113, "QU"):
//   diagonal   = diag(x)                      (extract_diagonal; x <- x - diag(diagonal))      QU:71-75
//   bucket[c]  = max_r |x[r, c]| / B          B = 127 (int8) or 32767 (int16)                  QU:85-86
//   code[r, c] = round_half_even(x[r, c] / (bucket[c] > 0 ? bucket[c] : 1))                    QU:89-94
//   to_float   = float(code) * bucket[c]  (+ diagonal on the diagonal)                         QU:108-112
// All of it is elementwise IEEE float32 arithmetic, so codes, diagonals and bucket sizes are
// BIT-EXACT with the reference; the only reduction is a max, which is order independent.
//
// HBM-bound byte work.  A tensor is viewed as [rows, cols] (rows = shape[0], the axis the
// reference reduces over) and cut into chunks of 64 rows x 256 columns; one 256-thread
// workgroup per chunk, for every tensor of the tree in ONE launch per pass:
//   pass 1  column max of |x| per chunk (float4 per lane: a wavefront reads 1 KiB of a row),
//           combined across chunks with atomicMax on the float bits (exact for non-negative
//           floats; a NaN has the largest bit pattern and therefore propagates like jnp.max);
//   pass 2  re-reads the chunk (L2 / Infinity-Cache resident for typical statistics), divides,
//           rounds to nearest even and stores 8 B (int16) or 4 B (int8) per lane.
// Algorithmic bytes per element: 4 (read) + 2 or 1 (write) for quantize (the second read is the
// price of the column-max dependency), 2 or 1 + 4 for dequantize.
#include <vector>

#include "common.h"
#include "options.h"

namespace psk {

#define PS_GLOBAL __attribute__((address_space(1)))

constexpr int QW = 256;  // columns per chunk
constexpr int QR = 64;   // rows per chunk

struct QTensor {
  const float* fin;   // quantize input
  float* fout;        // dequantize output
  void* codes;
  float* diag;
  float* bucket;
  unsigned* colmax;   // workspace: bit patterns of the column maxima (quantize only)
  long long rows, cols, ld, ldq;
  int bits, extract, vec4;
  int chunk0, strips;
  int flat;               // contiguous [rows, cols] float4-addressable tensor: the flat kernels take it
  int fchunk0, fchunks;   // its chunks of QFLAT consecutive elements
  int strip;              // quantize only: the register-resident strip kernel takes it (quant_strip_kernel)
  int schunk0, schunks;   // its strips of QS_COLS columns
  int small;              // quantize only: workgroups at the head of the strip launch take it (quant_small_body):
  int small_w;            //   every one ALL rows of a range of small_w columns
  int team;               // strips of more than 1024 rows: parts of 1024 rows, one workgroup each (1 = whole strips)
  unsigned* sarrive;      // workspace: per strip, the parts whose column maxima are merged
};

typedef float qf4 __attribute__((ext_vector_type(4)));
struct F4 { float x, y, z, w; };
__device__ inline F4 ldg4(const float* p) {
  const qf4 v = *(const qf4 PS_GLOBAL*)(p);
  return {v[0], v[1], v[2], v[3]};
}
__device__ inline float ldg1(const float* p) { return *(const float PS_GLOBAL*)(p); }
// (wave-uniform 64-bit base) + (one 32-bit byte offset per lane): the uniform part stays in SGPRs, so
// sixteen loads in flight cost ONE address VGPR instead of a 64-bit pair each (with per-load 64-bit
// addresses these kernels took 240-316 VGPRs: one wavefront per SIMD, slower than four loads in flight)
__device__ inline F4 ldg4_so(const void* ubase, uint32_t lane_off) {
  const qf4 v = *(const qf4 PS_GLOBAL*)((const char PS_GLOBAL*)ubase + (uint64_t)lane_off);
  return {v[0], v[1], v[2], v[3]};
}
__device__ inline uint2 ldg2u_so(const void* ubase, uint32_t lane_off) {
  typedef unsigned u2v __attribute__((ext_vector_type(2)));
  const u2v v = *(const u2v PS_GLOBAL*)((const char PS_GLOBAL*)ubase + (uint64_t)lane_off);
  return uint2{v[0], v[1]};
}
__device__ inline unsigned ldg1u_so(const void* ubase, uint32_t lane_off) {
  return *(const unsigned PS_GLOBAL*)((const char PS_GLOBAL*)ubase + (uint64_t)lane_off);
}

// ---- pass 1: column maxima -------------------------------------------------------------
__global__ __launch_bounds__(256) void quant_colmax_kernel(const QTensor* ts, const int* cmap, int chunk_base) {
  __shared__ unsigned red[4][QW];
  const int chunk = blockIdx.x + chunk_base;
  const QTensor* t = &ts[cmap[chunk]];
  if (t->flat) return;   // quant_flat_kernel took it
  const int local = chunk - t->chunk0;
  const int strip = local % t->strips, rc = local / t->strips;
  const int tid = threadIdx.x;
  const long long rows = t->rows, cols = t->cols, ld = t->ld;
  const long long r0 = (long long)rc * QR;
  const long long r1 = r0 + QR < rows ? r0 + QR : rows;
  if (t->vec4) {
    const int wave = tid >> 6, lane = tid & 63;
    const long long col = (long long)strip * QW + lane * 4;
    unsigned m[4] = {0u, 0u, 0u, 0u};
    if (col < cols) {
      // all 16 rows of the lane in flight (independent 16-byte loads) before the first use: with four
      // the kernel ran at 3.7 TB/s (latency-bound), the same access shape streams at 6 TB/s
      // (tools/bench_symv.hip v1)
      uint32_t loff = (uint32_t)((wave * ld + lane * 4) * 4);
      asm volatile("" : "+v"(loff));
      for (long long rb = r0 + wave; rb == r0 + wave; rb += 64) {   // QR = 64 rows: one pass
        F4 v[16];
        const float* ub = t->fin + (rb - wave) * ld + (long long)strip * QW;   // wave-uniform
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const long long r = rb + 4 * u;
          v[u] = r < r1 ? ldg4_so(ub + (long long)(4 * u) * ld, loff) : F4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const long long r = rb + 4 * u;
          float x[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (t->extract && col + j == r) x[j] = __fsub_rn(x[j], x[j]);   // QU:79-80 (NaN for a non-finite diagonal)
            const unsigned b = __float_as_uint(fabsf(x[j]));
            m[j] = b > m[j] ? b : m[j];
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[wave][lane * 4 + j] = m[j];
    __syncthreads();
    const long long c = (long long)strip * QW + tid;
    if (c < cols) {
      unsigned a = red[0][tid], b = red[1][tid], cc = red[2][tid], d = red[3][tid];
      a = a > b ? a : b; cc = cc > d ? cc : d; a = a > cc ? a : cc;
      if (a) atomicMax(t->colmax + c, a);
    }
  } else {
    // one column per lane; eight rows in flight (descriptor fields in locals: a store through another pointer
    // makes the compiler re-read them, which chained every row's load behind the one before)
    const long long c = (long long)strip * QW + tid;
    if (c >= cols) return;
    const float* const fin = t->fin;
    const int extract = t->extract;
    unsigned m = 0u;
    for (long long rb = r0; rb < r1; rb += 8) {
      float x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = rb + u < r1 ? ldg1(fin + (rb + u) * ld + c) : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (extract && c == rb + u) x[u] = __fsub_rn(x[u], x[u]);
        const unsigned b = __float_as_uint(fabsf(x[u]));
        m = b > m ? b : m;
      }
    }
    if (m) atomicMax(t->colmax + c, m);
  }
}

// ---- flat kernels (round 5): chunks of 16384 CONSECUTIVE elements ---------------------------------------
// A contiguous [rows, cols] tensor is streamed in runs of 64 KB (whole rows, back to back) instead of
// 64-row x 1 KB tiles: the 1 KB row pieces of a tile land in different DRAM pages (measured: the tile
// form decodes at 2.5 TB/s).  Column of an element = its flat offset mod cols, tracked incrementally;
// the column maxima of a chunk are merged in LDS (ds_max_u32) and flushed with one global atomicMax
// per column.  Same arithmetic as the tile kernels, bit for bit.
constexpr int QFLAT = 16384;      // elements per chunk (16 float4 per thread)
constexpr int QFLAT_MAXC = 8192;  // columns kept in LDS by the column-max pass
constexpr int QMAXT = 1 << 30;
__device__ inline int encode1(float x, float bs_nonzero);

// FULL: the chunk lies inside the tensor (every chunk but a tensor's last): no per-lane bounds tests.
// With them every load sits behind a divergent branch and the compiler waits for it before the next one:
// the sixteen loads of a lane are then issued one after the other (measured: half the rate of the same
// stream without the tests, tools/bench_stream.hip).
template <int MODE, bool FULL, int BITS>
__device__ __forceinline__ void quant_flat_body(const QTensor* t, int local, unsigned* s_max) {
  const int tid = threadIdx.x;
  const long long cols = t->cols, rows = t->rows, total = rows * cols;
  const long long base = (long long)local * QFLAT;
  const int extract = t->extract;
  const float nb = BITS == 8 ? 127.f : 32767.f;
  // incremental (row, col) of the thread's float4s: offsets base + 4 tid + 1024 k
  const long long off0 = base + 4 * tid;
  int row = (int)(off0 / cols);                      // rows * cols <= QMAXT: 32-bit row / column arithmetic
  int col = (int)(off0 - row * cols);
  const int step_c = (int)(1024 % cols);
  const int step_r = (int)(1024 / cols);
  uint32_t loff = (uint32_t)(4 * tid * 4);
  asm volatile("" : "+v"(loff));
  const char* fbase = reinterpret_cast<const char*>(t->fin) + base * 4;
  constexpr int esz = BITS == 16 ? 2 : 1;
  const char* cbase = reinterpret_cast<const char*>(t->codes) + base * esz;
  uint32_t coff = (uint32_t)(4 * tid * esz);
  asm volatile("" : "+v"(coff));
  F4 v[16];
  uint2 pk16[16];
  unsigned pk8[16];
  float dg[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const bool in = FULL || off0 + 1024 * k < total;
    if (MODE != 2) {
      v[k] = F4{0.f, 0.f, 0.f, 0.f};
      if (in) v[k] = ldg4_so(fbase + (long long)k * 4096, loff);
    } else {
      pk16[k] = uint2{0u, 0u}; pk8[k] = 0u;
      if (in) {
        if (BITS == 16) pk16[k] = ldg2u_so(cbase + (long long)k * 2048, coff);
        else pk8[k] = ldg1u_so(cbase + (long long)k * 1024, coff);
      }
    }
  }
  if (MODE == 2 && extract) {
    // the diagonal entry of the lane's row, requested with the codes by the lanes whose float4 holds the diagonal
    // element (a float4 never straddles a row: row - col in 0 .. 3)
    int prow = row, pcol = col;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      dg[k] = 0.f;
      if ((unsigned)(prow - pcol) < 4u && prow < (int)rows) dg[k] = ldg1(t->diag + prow);
      pcol += step_c; prow += step_r;
      if (pcol >= cols) { pcol -= (int)cols; prow += 1; }
    }
  }
  if (MODE == 1 && local == 0) {   // bucket sizes (QU:86), once per tensor
    for (int c = tid; c < cols; c += 256) t->bucket[c] = __fdiv_rn(__uint_as_float(t->colmax[c]), nb);
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const bool in = FULL || off0 + 1024 * k < total;
    if (in) {
      if (MODE == 0) {
        float x[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (extract && col + j == row) x[j] = __fsub_rn(x[j], x[j]);
          const unsigned b = __float_as_uint(fabsf(x[j]));
          if (b) atomicMax(&s_max[col + j], b);
        }
      } else if (MODE == 1) {
        float x[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
        typedef unsigned u4v __attribute__((ext_vector_type(4)));
        const u4v cm = *(const u4v PS_GLOBAL*)(t->colmax + col);
        const unsigned cmv[4] = {cm[0], cm[1], cm[2], cm[3]};
        int q[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float bs = __fdiv_rn(__uint_as_float(cmv[j]), nb);   // QU:86
          const float bnz = bs > 0.f ? bs : 1.f;                       // QU:89-90
          if (extract && col + j == row) {
            *(float PS_GLOBAL*)(t->diag + row) = x[j];
            x[j] = __fsub_rn(x[j], x[j]);
          }
          q[j] = encode1(x[j], bnz);
        }
        const long long o = off0 + 1024 * k;
        if (BITS == 16) {
          typedef unsigned u2v __attribute__((ext_vector_type(2)));
          u2v pk;
          pk[0] = ((unsigned)q[0] & 0xffffu) | ((unsigned)q[1] << 16);
          pk[1] = ((unsigned)q[2] & 0xffffu) | ((unsigned)q[3] << 16);
          *(u2v PS_GLOBAL*)(reinterpret_cast<short*>(t->codes) + o) = pk;
        } else {
          const unsigned pk = ((unsigned)q[0] & 0xffu) | (((unsigned)q[1] & 0xffu) << 8) |
                              (((unsigned)q[2] & 0xffu) << 16) | ((unsigned)q[3] << 24);
          *(unsigned PS_GLOBAL*)(reinterpret_cast<signed char*>(t->codes) + o) = pk;
        }
      } else {
        int q[4];
        if (BITS == 16) {
          q[0] = (short)(pk16[k].x & 0xffffu); q[1] = (short)(pk16[k].x >> 16);
          q[2] = (short)(pk16[k].y & 0xffffu); q[3] = (short)(pk16[k].y >> 16);
        } else {
          const unsigned pk = pk8[k];
          q[0] = (signed char)(pk & 0xffu); q[1] = (signed char)((pk >> 8) & 0xffu);
          q[2] = (signed char)((pk >> 16) & 0xffu); q[3] = (signed char)(pk >> 24);
        }
        const F4 b4 = ldg4(t->bucket + col);
        const float bs[4] = {b4.x, b4.y, b4.z, b4.w};
        qf4 x;
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = __fmul_rn((float)q[j], bs[j]);          // QU:109
        const int dj = row - col;
        if (extract && (unsigned)dj < 4u) {                                         // QU:111
#pragma unroll
          for (int j = 0; j < 4; ++j) x[j] = dj == j ? __fadd_rn(x[j], dg[k]) : x[j];
        }
        *(qf4 PS_GLOBAL*)(t->fout + off0 + 1024 * k) = x;
      }
    }
    col += step_c; row += step_r;
    if (col >= cols) { col -= (int)cols; row += 1; }
  }
}

template <int MODE>   // 0 = column maxima, 1 = encode, 2 = decode
__global__ __launch_bounds__(256) void quant_flat_kernel(const QTensor* ts, const int* cmap) {
  __shared__ unsigned s_max[MODE == 0 ? QFLAT_MAXC : 1];
  // chunk -> tensor from a per-chunk table (one 4-byte load)
  const QTensor* t = &ts[cmap[blockIdx.x]];
  const int local = blockIdx.x - t->fchunk0;
  const int tid = threadIdx.x;
  const long long cols = t->cols;
  if (MODE == 0) {
    for (int c = tid; c < cols; c += 256) s_max[c] = 0u;
    __syncthreads();
  }
  const bool full = (long long)(local + 1) * QFLAT <= t->rows * cols;
  // (MODE 0 does not depend on the code width)
  if (MODE == 0 || t->bits == 16) {
    if (full) quant_flat_body<MODE, true, 16>(t, local, s_max);
    else quant_flat_body<MODE, false, 16>(t, local, s_max);
  } else {
    if (full) quant_flat_body<MODE, true, 8>(t, local, s_max);
    else quant_flat_body<MODE, false, 8>(t, local, s_max);
  }
  if (MODE == 0) {
    __syncthreads();
    for (int c = tid; c < cols; c += 256) {
      const unsigned a = s_max[c];
      if (a) atomicMax(t->colmax + c, a);
    }
  }
}

__device__ inline int encode1(float x, float bs_nonzero) {
  return (int)rintf(__fdiv_rn(x, bs_nonzero));  // QU:91-94; rint = round half to even
}

// ---- quantize in ONE read (round 6): register-resident column strips -----------------------------------------
// The scale of a column is the maximum over ALL its rows, so the streaming kernels above read the input twice
// (column maxima, then codes: 4 + 4 + 2 bytes of traffic per element for int16).  For matrices of up to 1024 rows a
// workgroup of 512 threads can keep a strip of 64 columns x all rows in REGISTERS (32 float4 per lane): one read,
// maxima through the wavefronts' lanes and 2 KB of LDS, codes from the same registers: 4 + 2 bytes per element.
// Lane = (column quad q = tid & 15, row group g = tid >> 4): a wavefront reads 4 rows x 256 contiguous bytes per
// instruction and has all of a lane's loads in flight at once (<= 32 x 16 bytes per lane, 256 KB per workgroup); int16
// codes leave as whole 128-byte lines.  Same arithmetic as the other kernels (IEEE division, round half to even, the
// maximum taken on bit patterns): codes, diagonals and bucket sizes are bit-identical.
constexpr int QS_COLS = 64;
constexpr int QS_THREADS = 512;
constexpr int QS_GROUPS = QS_THREADS / 16;   // 32 row groups
constexpr int QS_NV = 32;                    // rows per lane: up to 32 x 32 = 1024 rows
constexpr int QS_MINR = 64;                  // below that a strip is too little work per workgroup

// Instruction count matters as much as bytes here: a workgroup's 65536 elements pass through the VALU twice (maximum,
// codes) behind loads that cannot overlap them (one workgroup per CU).  The first version spent ~45 instructions per
// element (the compiler's 11-instruction IEEE division, 64-bit row/column tests per element, spilled scalars) and was
// VALU-bound at 0.37 of the HBM peak.  Here:
//  * the diagonal (extract) is handled before the passes, for the two steps k whose rows can cross the strip's
//    columns (a scalar test per step instead of a 64-bit comparison per element);
//  * the division x / b runs the SAME operation sequence as the compiler's expansion of IEEE division
//    (rcp, two fmas refining it, then mul, fma, fma, fma, fma) with the part that depends only on the column's b
//    hoisted out of the element loop: five instructions per element.  The expansion's scaling (v_div_scale) and
//    fix-up (v_div_fixup) steps leave operands and result unchanged unless the denominator is outside
//    [2^-60, 2^60] (kept out of this path) or the quotient is below 2^-43 (rounds to code 0 either way), so the codes
//    are those of __fdiv_rn bit for bit; a wavefront with any column outside the range takes the plain division;
//  * round-half-even + integer conversion + packing: q + 1.5*2^23 holds the rounded integer in its low mantissa
//    bits (|q| <= 32767), two byte-permutes pack four of them.
template <int BITS>
__device__ __forceinline__ unsigned qs_pack_pair(unsigned lo_bits, unsigned hi_bits) {
  // low 16 (or 8) bits of each operand, concatenated
  return BITS == 16 ? __builtin_amdgcn_perm(hi_bits, lo_bits, 0x05040100u)
                    : __builtin_amdgcn_perm(hi_bits, lo_bits, 0x0c0c0400u);
}

// One workgroup per CU (152+ VGPRs x 8 wavefronts): within ONE strip the loads cannot overlap the codes' stores, and
// a second workgroup does not fit.  So the kernel is persistent -- 256 workgroups (one per CU) take strips from a queue
// (an atomic counter) -- and rolls the register file: step k of the encode pass stores the codes of row group k and
// at once loads row group k of the NEXT strip into the registers it has just freed.  The memory pipe then idles only
// during the maximum pass and its reduction.
struct QsStrip {          // wave-uniform description of one strip, or of one part of a tall strip (scalars)
  const char* lb;         // first row of the wavefront's row groups, first column of the strip (input)
  char* cb;               // same, codes
  float* diag; float* bucket;
  unsigned* colmax;       // team > 1: the tensor's merged column maxima (bit patterns), the strip's arrival counter
  unsigned* arrive;
  int rows, cols, c0, extract, bits;   // rows: of this part
  int r_off, team;        // first row of the part; parts per strip
};

// (after the first store the compiler reads descriptor fields through the vector unit: back to scalars)
__device__ __forceinline__ int qs_uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
template <class P>
__device__ __forceinline__ P* qs_uni(P* p) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
  return reinterpret_cast<P*>(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ QsStrip qs_strip(const QTensor* ts, const int* cmap, int idx, int wave) {
  const QTensor* t = &ts[qs_uni(cmap[idx])];
  QsStrip s;
  const int local = idx - qs_uni(t->schunk0);
  s.team = qs_uni(t->team);
  const int strip = local / s.team, part = local - strip * s.team;
  const int rows_all = qs_uni((int)t->rows);                        // rows_all * cols * 4 < 2^31
  s.r_off = part * (QS_NV * QS_GROUPS);
  s.rows = rows_all - s.r_off < QS_NV * QS_GROUPS ? rows_all - s.r_off : QS_NV * QS_GROUPS;
  s.cols = qs_uni((int)t->cols);
  s.c0 = strip * QS_COLS;
  s.extract = qs_uni(t->extract); s.bits = qs_uni(t->bits);
  s.diag = qs_uni(t->diag); s.bucket = qs_uni(t->bucket);
  s.colmax = qs_uni(t->colmax);
  s.arrive = s.team > 1 ? qs_uni(t->sarrive) + strip : nullptr;
  s.lb = reinterpret_cast<const char*>(qs_uni(t->fin) + (long long)(s.r_off + 4 * wave) * s.cols + s.c0);
  s.cb = reinterpret_cast<char*>(qs_uni(t->codes)) +
         ((long long)(s.r_off + 4 * wave) * s.cols + s.c0) * (s.bits == 16 ? 2 : 1);
  return s;
}

// Small tensors (up to QSM_ELEMS elements, any alignment and stride: biases, a 197 x 197 statistic, vectors) ride in
// the strip launch, one workgroup each, FIRST in the grid: column maxima through LDS, then the codes from a second
// read (L2).  As launches of their own (tile kernels: two passes, two launches) they ran 20-50 us behind the big kernel
// on the same stream.  Same arithmetic as every other path: bit-identical codes.
constexpr int QSM_ELEMS = 65536;
constexpr int QSM_COLS = 2048;

// item = (tensor, first column): ALL rows of a range of up to `width` columns (QTensor::small_w; rows * width <=
// QSM_ELEMS, width <= QSM_COLS).  Columns are independent, so a wide and short tensor ([1, 197, 768] embeddings:
// 151 296 columns of one row) is simply many items.
__device__ __forceinline__ void quant_small_body(const QTensor* t, int c0, unsigned* s_max) {
  const int tid = threadIdx.x;
  const int rows = (int)t->rows, cols = (int)t->cols;
  const int nc = cols - c0 < t->small_w ? cols - c0 : t->small_w, total = rows * nc;
  const long long ld = t->ld, ldq = t->ldq;
  const float* const fin = t->fin;
  void* const codes = t->codes;
  float* const diag = t->diag;
  float* const bucket = t->bucket;
  const int extract = t->extract, bits = t->bits;
  const float nb = bits == 8 ? 127.f : 32767.f;
  for (int c = tid; c < nc; c += QS_THREADS) s_max[c] = 0u;
  __syncthreads();
  // pass 1: a lane keeps the running maximum of the column it is in and merges it when the column changes
  // (vectors, and widths that divide the thread count, merge once per lane)
  {
    int ccol = -1;
    unsigned cmax = 0u;
    for (int e0 = tid; e0 < total; e0 += 8 * QS_THREADS) {
      float x[8];
      int rr[8], cc[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + u * QS_THREADS;
        rr[u] = e / nc; cc[u] = e - rr[u] * nc;
        x[u] = e < total ? ldg1(fin + (long long)rr[u] * ld + c0 + cc[u]) : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (e0 + u * QS_THREADS >= total) break;
        if (extract && rr[u] == c0 + cc[u]) x[u] = __fsub_rn(x[u], x[u]);   // QU:79-80
        const unsigned b = __float_as_uint(x[u]) & 0x7fffffffu;
        if (cc[u] != ccol) {
          if (cmax) atomicMax(&s_max[ccol], cmax);
          ccol = cc[u]; cmax = 0u;
        }
        cmax = b > cmax ? b : cmax;
      }
    }
    if (cmax) atomicMax(&s_max[ccol], cmax);
  }
  __syncthreads();
  for (int c = tid; c < nc; c += QS_THREADS) {
    const float bs = __fdiv_rn(__uint_as_float(s_max[c]), nb);   // QU:86
    bucket[c0 + c] = bs;
    s_max[c] = __float_as_uint(bs > 0.f ? bs : 1.f);             // QU:89-90
  }
  __syncthreads();
  for (int e0 = tid; e0 < total; e0 += 8 * QS_THREADS) {
    float x[8];
    int rr[8], cc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + u * QS_THREADS;
      rr[u] = e / nc; cc[u] = e - rr[u] * nc;
      x[u] = e < total ? ldg1(fin + (long long)rr[u] * ld + c0 + cc[u]) : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (e0 + u * QS_THREADS >= total) break;
      if (extract && rr[u] == c0 + cc[u]) { diag[rr[u]] = x[u]; x[u] = __fsub_rn(x[u], x[u]); }
      const int q = encode1(x[u], __uint_as_float(s_max[cc[u]]));
      if (bits == 16) reinterpret_cast<short*>(codes)[(long long)rr[u] * ldq + c0 + cc[u]] = (short)q;
      else reinterpret_cast<signed char*>(codes)[(long long)rr[u] * ldq + c0 + cc[u]] = (signed char)q;
    }
  }
}

__global__ __launch_bounds__(QS_THREADS) void quant_strip_kernel(const QTensor* ts, const int* cmap, int nstrips,
                                                                unsigned* queue, const int* smap, int nsmall) {
  __shared__ unsigned s_mem[QSM_COLS];   // the strips' reduction buffers (2 x 8 x 64), a small tensor's column maxima
  __shared__ int s_next[2];
  __shared__ int s_team_ok;
  if ((int)blockIdx.x < nsmall) {
    quant_small_body(&ts[smap[2 * blockIdx.x]], smap[2 * blockIdx.x + 1], s_mem);
    return;
  }
  unsigned (*s_red)[QS_THREADS / 64][QS_COLS] = reinterpret_cast<unsigned (*)[QS_THREADS / 64][QS_COLS]>(s_mem);
  const int tid = threadIdx.x, q = tid & 15, g = tid >> 4, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // every item comes from the queue, the first one too: only RUNNING workgroups hold items (a part of a tall strip
  // waits for its team mates -- see below -- and a workgroup that is not resident yet must not be one of them)
  if (tid == 0) s_next[1] = (int)__hip_atomic_fetch_add(queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const int idx = __builtin_amdgcn_readfirstlane(s_next[1]);
  if (idx >= nstrips) return;
  QsStrip cur = qs_strip(ts, cmap, idx, wave);
  F4 v[QS_NV];
  {
    // the first strip's loads (every later strip's are issued by the encode pass of the strip before it)
    const bool col_in = cur.c0 + 4 * q < cur.cols;         // cols % 4 == 0: a quad is inside or outside as a whole
    uint32_t loff = (uint32_t)((((g & 3) * cur.cols) + 4 * q) * 4);
    asm volatile("" : "+v"(loff));
    const long long step = (long long)QS_GROUPS * cur.cols * 4;
    const char* lb = qs_uni(cur.lb);
#pragma unroll
    for (int k = 0; k < QS_NV; ++k) {
      v[k] = F4{0.f, 0.f, 0.f, 0.f};
      if (col_in && QS_GROUPS * k + g < cur.rows) v[k] = ldg4_so(lb, loff);
      lb += step;
      asm volatile("" : "+s"(lb));
    }
  }
  for (int it = 0;; ++it) {
    const int par = it & 1;
    // (a part of a tall strip takes its next item only AFTER its team has met: see the team step)
    if (tid == 0 && cur.team == 1)
      s_next[par] = (int)__hip_atomic_fetch_add(queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int rows = cur.rows, cols = cur.cols;
    const int col = cur.c0 + 4 * q;
    const bool col_in = col < cols;
    const float nb = cur.bits == 8 ? 127.f : 32767.f;
    const int nv_used = (rows + QS_GROUPS - 1) / QS_GROUPS;
    // diagonal: row 32 k + g meets the strip's columns [c0, c0 + 64) only for k = c0 / 32 and k = c0 / 32 + 1
    if (cur.extract) {
      const int kd = (cur.c0 - cur.r_off) / QS_GROUPS;     // (exact: both are multiples of 32; may lie outside 0 .. 31)
#pragma unroll
      for (int k = 0; k < QS_NV; ++k) {
        if (k != kd && k != kd + 1) continue;              // scalar
        const int r = QS_GROUPS * k + g;
        const int d = cur.r_off + r - col;                 // the lane holds the diagonal element in component d
        if ((unsigned)d < 4u && r < rows && col_in) {
          float* xs = &v[k].x;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (d == j) {
              *(float PS_GLOBAL*)(cur.diag + cur.r_off + r) = xs[j];
              xs[j] = __fsub_rn(xs[j], xs[j]);             // QU:79-80: value - diag(diagonal)
            }
          }
        }
      }
    }
    unsigned m[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int k = 0; k < QS_NV; ++k) {
      if (k >= nv_used) continue;                          // scalar; (rows not loaded are zeros)
      const float x[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned b = __float_as_uint(x[j]) & 0x7fffffffu;
        m[j] = b > m[j] ? b : m[j];
      }
    }
    // the four row groups of a wavefront (lanes q, q + 16, q + 32, q + 48), then the eight wavefronts through LDS
    // (two buffers: a wavefront may enter the next strip's reduction while another still reads this one's)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned o = __shfl_xor(m[j], 16, 64); m[j] = o > m[j] ? o : m[j];
      o = __shfl_xor(m[j], 32, 64); m[j] = o > m[j] ? o : m[j];
    }
    if (lane < 16) {
#pragma unroll
      for (int j = 0; j < 4; ++j) s_red[par][wave][4 * q + j] = m[j];
    }
    __syncthreads();
    unsigned amax[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned a = s_red[par][0][4 * q + j];
#pragma unroll
      for (int w = 1; w < QS_THREADS / 64; ++w) { const unsigned o = s_red[par][w][4 * q + j]; a = o > a ? o : a; }
      amax[j] = a;
    }
    if (cur.team > 1) {                                      // scalar
      // Team step of a tall strip (1024 < rows <= 4096): this workgroup holds 1024 rows of it.  The parts merge their
      // maxima with agent-scope atomics (64 per part), count themselves in and wait for the others.  No deadlock: items
      // leave the queue in order, a strip's parts are consecutive items, only running workgroups hold items and a
      // waiting one holds exactly one (it takes its next item after the wait): every team but the one at the queue's
      // head is complete among the running workgroups, and that one is completed by the next workgroup to finish
      // an item or to start.  The wait is bounded (10 s); on expiry the part publishes NaN bucket sizes.
      // The merges must be PERFORMED before this part counts itself in.  They are returning atomics whose results are
      // consumed (the wave waits for the values, which come back from the point of coherence); a workgroup-scope
      // release fence is not enough -- outside tgsplit mode the compiler emits no vmcnt wait for it, and the arrival
      // (another address, another channel) could become visible first: a team mate then read a stale maximum
      // (seen as one wrong bucket size in ~30 runs of the tall-matrix tests).
      if (g == 0 && col_in) {
        unsigned seen = 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (amax[j])
            seen |= __hip_atomic_fetch_max(cur.colmax + col + j, amax[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("" :: "v"(seen));
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        __hip_atomic_fetch_add(cur.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long deadline = __builtin_amdgcn_s_memrealtime() + 1000000000ull;   // 100 MHz clock
        int ok = 1;
        while (__hip_atomic_load(cur.arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)cur.team) {
          if (__builtin_amdgcn_s_memrealtime() > deadline) { ok = 0; break; }
          __builtin_amdgcn_s_sleep(2);
        }
        s_next[par] = (int)__hip_atomic_fetch_add(queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_team_ok = ok;
      }
      __syncthreads();
      const bool expired = s_team_ok == 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        amax[j] = col_in ? __hip_atomic_load(cur.colmax + col + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        if (expired) amax[j] = 0x7fc00000u;
      }
    }
    float bnz[4], y1[4];
    bool sane = true;
    const bool writes_bucket = g == 0 && col_in && cur.r_off == 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float bs = __fdiv_rn(__uint_as_float(amax[j]), nb);   // QU:86
      bnz[j] = bs > 0.f ? bs : 1.f;                          // QU:89-90
      if (writes_bucket) *(float PS_GLOBAL*)(cur.bucket + col + j) = bs;
      // the denominator-only part of the IEEE division sequence
      const float rc = __builtin_amdgcn_rcpf(bnz[j]);
      const float e = __fmaf_rn(-bnz[j], rc, 1.0f);
      y1[j] = __fmaf_rn(e, rc, rc);
      sane = sane && bnz[j] >= 0x1p-60f && bnz[j] <= 0x1p60f;
    }
    const bool fast = __all(sane ? 1 : 0) != 0;             // wave-uniform
    // the next strip (wave-uniform: read through the scalar unit)
    const int nxt = __builtin_amdgcn_readfirstlane(s_next[par]);
    const bool has_next = nxt < nstrips;
    QsStrip nx = cur;
    if (has_next) nx = qs_strip(ts, cmap, nxt, wave);
    const bool ncol_in = has_next && nx.c0 + 4 * q < nx.cols;
    uint32_t nloff = (uint32_t)((((g & 3) * nx.cols) + 4 * q) * 4);
    asm volatile("" : "+v"(nloff));
    const long long nstep = (long long)QS_GROUPS * nx.cols * 4;
    const char* nlb = qs_uni(nx.lb);
    const bool b16 = cur.bits == 16;
    char* cb = qs_uni(cur.cb);
    uint32_t coff = (uint32_t)((((g & 3) * cols) + 4 * q) * (b16 ? 2 : 1));
    asm volatile("" : "+v"(coff));
    const long long cstep = (long long)QS_GROUPS * cols * (b16 ? 2 : 1);
#pragma unroll
    for (int k = 0; k < QS_NV; ++k) {
      if (k < nv_used) {                                     // scalar
        const float x[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
        unsigned cq[4];
        if (fast) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float q0 = __fmul_rn(x[j], y1[j]);
            const float r0 = __fmaf_rn(-bnz[j], q0, x[j]);
            const float q1 = __fmaf_rn(r0, y1[j], q0);
            const float r1 = __fmaf_rn(-bnz[j], q1, x[j]);
            const float qq = __fmaf_rn(r1, y1[j], q1);
            cq[j] = __float_as_uint(__fadd_rn(qq, 12582912.0f));   // 1.5 * 2^23: integer in the low mantissa bits
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) cq[j] = (unsigned)encode1(x[j], bnz[j]);
        }
        const bool valid = col_in && QS_GROUPS * k + g < rows;
        char* dst = cb + coff;
        if (b16) {                                           // scalar
          typedef unsigned u2v __attribute__((ext_vector_type(2)));
          u2v pk;
          pk[0] = qs_pack_pair<16>(cq[0], cq[1]);
          pk[1] = qs_pack_pair<16>(cq[2], cq[3]);
          if (valid) *(u2v PS_GLOBAL*)dst = pk;
        } else {
          const unsigned lo = qs_pack_pair<8>(cq[0], cq[1]), hi = qs_pack_pair<8>(cq[2], cq[3]);
          const unsigned pk = __builtin_amdgcn_perm(hi, lo, 0x05040100u);
          if (valid) *(unsigned PS_GLOBAL*)dst = pk;
        }
      }
      cb += cstep;
      asm volatile("" : "+s"(cb));
      // row group k of the next strip into the registers just encoded
      v[k] = F4{0.f, 0.f, 0.f, 0.f};
      if (ncol_in && QS_GROUPS * k + g < nx.rows) v[k] = ldg4_so(nlb, nloff);
      nlb += nstep;
      asm volatile("" : "+s"(nlb));
    }
    if (!has_next) break;
    cur = nx;
  }
}

// ---- pass 2: bucket sizes, diagonal, codes ---------------------------------------------
__global__ __launch_bounds__(256) void quant_encode_kernel(const QTensor* ts, const int* cmap, int chunk_base) {
  const int chunk = blockIdx.x + chunk_base;
  const QTensor* t = &ts[cmap[chunk]];
  if (t->flat) return;   // quant_flat_kernel took it
  const int local = chunk - t->chunk0;
  const int strip = local % t->strips, rc = local / t->strips;
  const int tid = threadIdx.x;
  const long long rows = t->rows, cols = t->cols, ld = t->ld, ldq = t->ldq;
  const long long r0 = (long long)rc * QR;
  const long long r1 = r0 + QR < rows ? r0 + QR : rows;
  const float nb = t->bits == 8 ? 127.f : 32767.f;
  if (t->vec4) {
    const int wave = tid >> 6, lane = tid & 63;
    const long long col = (long long)strip * QW + lane * 4;
    if (col >= cols) return;
    float bs[4], bnz[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bs[j] = __fdiv_rn(__uint_as_float(t->colmax[col + j]), nb);  // QU:86
      bnz[j] = bs[j] > 0.f ? bs[j] : 1.f;                          // QU:89-90
    }
    if (rc == 0 && wave == 0)
      *reinterpret_cast<float4*>(t->bucket + col) = make_float4(bs[0], bs[1], bs[2], bs[3]);
    uint32_t loff = (uint32_t)((wave * ld + lane * 4) * 4);
    asm volatile("" : "+v"(loff));
    for (long long rb = r0 + wave; rb == r0 + wave; rb += 64) {   // QR = 64 rows: one pass
      F4 v[16];
      const float* ub = t->fin + (rb - wave) * ld + (long long)strip * QW;   // wave-uniform
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const long long r = rb + 4 * u;
        v[u] = r < r1 ? ldg4_so(ub + (long long)(4 * u) * ld, loff) : F4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const long long r = rb + 4 * u;
        if (r >= r1) continue;   // (not `break`: the loop must unroll completely, or v[] lives in scratch)
        float x[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
        int q[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (t->extract && col + j == r) { t->diag[r] = x[j]; x[j] = __fsub_rn(x[j], x[j]); }
          q[j] = encode1(x[j], bnz[j]);
        }
        if (t->bits == 16) {
          uint2 pk;
          pk.x = ((unsigned)q[0] & 0xffffu) | ((unsigned)q[1] << 16);
          pk.y = ((unsigned)q[2] & 0xffffu) | ((unsigned)q[3] << 16);
          *reinterpret_cast<uint2*>(reinterpret_cast<short*>(t->codes) + r * ldq + col) = pk;
        } else {
          const unsigned pk = ((unsigned)q[0] & 0xffu) | (((unsigned)q[1] & 0xffu) << 8) |
                              (((unsigned)q[2] & 0xffu) << 16) | ((unsigned)q[3] << 24);
          *reinterpret_cast<unsigned*>(reinterpret_cast<signed char*>(t->codes) + r * ldq + col) = pk;
        }
      }
    }
  } else {
    const long long c = (long long)strip * QW + tid;
    if (c >= cols) return;
    const float bs = __fdiv_rn(__uint_as_float(t->colmax[c]), nb);
    const float bnz = bs > 0.f ? bs : 1.f;
    const float* const fin = t->fin;
    float* const diag = t->diag;
    void* const codes = t->codes;
    const int extract = t->extract, bits = t->bits;
    if (rc == 0) t->bucket[c] = bs;
    for (long long rb = r0; rb < r1; rb += 8) {
      float x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = rb + u < r1 ? ldg1(fin + (rb + u) * ld + c) : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const long long r = rb + u;
        if (r >= r1) break;
        if (extract && c == r) { diag[r] = x[u]; x[u] = __fsub_rn(x[u], x[u]); }
        const int q = encode1(x[u], bnz);
        if (bits == 16) reinterpret_cast<short*>(codes)[r * ldq + c] = (short)q;
        else reinterpret_cast<signed char*>(codes)[r * ldq + c] = (signed char)q;
      }
    }
  }
}

// ---- to_float --------------------------------------------------------------------------
__device__ __forceinline__ void quant_decode_tile(const QTensor* ts, const int* cmap, int chunk) {
  const QTensor* t = &ts[cmap[chunk]];
  if (t->flat) return;
  const int local = chunk - t->chunk0;
  const int strip = local % t->strips, rc = local / t->strips;
  const int tid = threadIdx.x;
  const long long rows = t->rows, cols = t->cols, ld = t->ld, ldq = t->ldq;
  const long long r0 = (long long)rc * QR;
  const long long r1 = r0 + QR < rows ? r0 + QR : rows;
  if (t->vec4) {
    const int wave = tid >> 6, lane = tid & 63;
    const long long col = (long long)strip * QW + lane * 4;
    if (col >= cols) return;
    const F4 b4 = ldg4(t->bucket + col);
    const float bs[4] = {b4.x, b4.y, b4.z, b4.w};
    // all 16 rows of the lane in flight before the first use (one load per lane at a time leaves the
    // kernel latency-bound: 3.2 TB/s of its 6 bytes per element)
    const int esz = t->bits == 16 ? 2 : 1;
    uint32_t loff = (uint32_t)((wave * ldq + lane * 4) * esz);
    asm volatile("" : "+v"(loff));
    for (long long rb = r0 + wave; rb == r0 + wave; rb += 64) {   // QR = 64 rows: one pass
      uint2 pk16[16];
      unsigned pk8[16];
      const char* ub = reinterpret_cast<const char*>(t->codes) + ((rb - wave) * ldq + (long long)strip * QW) * esz;
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const long long r = rb + 4 * u;
        pk16[u] = uint2{0u, 0u}; pk8[u] = 0u;
        if (r < r1) {
          if (t->bits == 16) pk16[u] = ldg2u_so(ub + (long long)(4 * u) * ldq * 2, loff);
          else pk8[u] = ldg1u_so(ub + (long long)(4 * u) * ldq, loff);
        }
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const long long r = rb + 4 * u;
        if (r >= r1) continue;   // (not `break`: the loop must unroll completely, or v[] lives in scratch)
        int q[4];
        if (t->bits == 16) {
          q[0] = (short)(pk16[u].x & 0xffffu); q[1] = (short)(pk16[u].x >> 16);
          q[2] = (short)(pk16[u].y & 0xffffu); q[3] = (short)(pk16[u].y >> 16);
        } else {
          const unsigned pk = pk8[u];
          q[0] = (signed char)(pk & 0xffu); q[1] = (signed char)((pk >> 8) & 0xffu);
          q[2] = (signed char)((pk >> 16) & 0xffu); q[3] = (signed char)(pk >> 24);
        }
        float x[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          x[j] = __fmul_rn((float)q[j], bs[j]);                                     // QU:109
          if (t->extract && col + j == r) x[j] = __fadd_rn(x[j], t->diag[r]);       // QU:111
        }
        *reinterpret_cast<float4*>(t->fout + r * ld + col) = make_float4(x[0], x[1], x[2], x[3]);
      }
    }
  } else {
    const long long c = (long long)strip * QW + tid;
    if (c >= cols) return;
    const float bs = t->bucket[c];
    const void* const codes = t->codes;
    const float* const diag = t->diag;
    float* const fout = t->fout;
    const int extract = t->extract, bits = t->bits;
    const float dg = (extract && c >= r0 && c < r1) ? diag[c] : 0.f;
    for (long long rb = r0; rb < r1; rb += 8) {
      int q[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const long long r = rb + u < r1 ? rb + u : r1 - 1;
        q[u] = bits == 16 ? (int)reinterpret_cast<const short*>(codes)[r * ldq + c]
                          : (int)reinterpret_cast<const signed char*>(codes)[r * ldq + c];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const long long r = rb + u;
        if (r >= r1) break;
        float x = __fmul_rn((float)q[u], bs);                   // QU:109
        if (extract && c == r) x = __fadd_rn(x, dg);            // QU:111
        fout[r * ld + c] = x;
      }
    }
  }
}

// to_float of a whole descriptor list in ONE launch: the tile chunks (strided views, tensors that are not
// float4-addressable -- few, latency-bound workgroups) come first in the grid and finish under the stream of flat
// chunks; as a launch of their own they ran 20-50 us BEHIND the big kernel on the same stream.
__global__ __launch_bounds__(256) void quant_decode_all_kernel(const QTensor* ts, const int* cmap, int fch, int tch) {
  if ((int)blockIdx.x < tch) {
    quant_decode_tile(ts, cmap + fch, (int)blockIdx.x);
    return;
  }
  const int chunk = (int)blockIdx.x - tch;
  const QTensor* t = &ts[cmap[chunk]];
  const int local = chunk - t->fchunk0;
  const bool full = (long long)(local + 1) * QFLAT <= t->rows * t->cols;
  if (t->bits == 16) {
    if (full) quant_flat_body<2, true, 16>(t, local, nullptr);
    else quant_flat_body<2, false, 16>(t, local, nullptr);
  } else {
    if (full) quant_flat_body<2, true, 8>(t, local, nullptr);
    else quant_flat_body<2, false, 8>(t, local, nullptr);
  }
}

static bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// Validates descriptors and fills the device table image.  Returns total chunks or <0.
static long long build_tensors(const ps_quant_desc* desc, int count, bool encode,
                               std::vector<QTensor>& ht, size_t& total_cols) {
  ht.resize(count);
  const bool flat_on = psh::resolve(nullptr).quant_flat != 0;
  const bool strip_on = psh::resolve(nullptr).quant_strip != 0;
  long long chunk0 = 0;
  total_cols = 0;
  for (int i = 0; i < count; ++i) {
    const ps_quant_desc& d = desc[i];
    if (d.rows < 0 || d.cols < 0 || (d.bits != 8 && d.bits != 16)) return PS_EINVAL;
    if (d.rows > 0 && d.cols > 0) {
      if (!d.fvalue || !d.codes || !d.bucket_size || d.ld < d.cols || d.ldq < d.cols)
        return PS_EINVAL;
      if (d.extract_diagonal && (!d.diagonal || d.rows != d.cols)) return PS_EINVAL;  // QU:67-69
    }
    QTensor& t = ht[i];
    t.fin = d.fvalue; t.fout = d.fvalue; t.codes = d.codes; t.diag = d.diagonal;
    t.bucket = d.bucket_size; t.colmax = nullptr;
    t.rows = d.rows; t.cols = d.cols; t.ld = d.ld; t.ldq = d.ldq;
    t.bits = d.bits; t.extract = d.extract_diagonal ? 1 : 0;
    const size_t code_vec = d.bits == 16 ? 8 : 4;
    t.vec4 = (d.cols % 4 == 0 && d.ld % 4 == 0 && d.ldq % 4 == 0 && aligned(d.fvalue, 16) &&
              aligned(d.codes, code_vec) && aligned(d.bucket_size, 16)) ? 1 : 0;
    t.flat = (flat_on && t.vec4 && d.ld == d.cols && d.ldq == d.cols && d.cols <= QFLAT_MAXC && d.rows > 0 &&
              d.cols > 0 && count <= QMAXT && aligned(d.codes, 16)) ? 1 : 0;
    // quantize: matrices of 64 ... 1024 rows are encoded from registers in one read (quant_strip_kernel),
    // (taller ones, up to 4096 rows, in parts of 1024 rows that merge their column maxima: QTensor::team)
    t.strip = (encode && strip_on && t.flat && d.rows >= QS_MINR && d.rows <= 4LL * QS_NV * QS_GROUPS &&
               (d.rows * d.cols) * 4 < (1LL << 31)) ? 1 : 0;
    t.team = t.strip ? (int)((d.rows + QS_NV * QS_GROUPS - 1) / (QS_NV * QS_GROUPS)) : 1;
    t.sarrive = nullptr;
    t.schunks = t.strip ? (int)((d.cols + QS_COLS - 1) / QS_COLS) * t.team : 0;
    // ... and small ones of any layout by one workgroup of the same launch (quant_small_body)
    // (short ones, rows < 64, of any width; others -- odd sizes, strided views -- up to 1024 rows)
    t.small = (encode && strip_on && !t.strip && d.rows > 0 && d.cols > 0 && d.rows <= QS_NV * QS_GROUPS) ? 1 : 0;
    t.small_w = t.small ? (int)std::min<long long>(std::min<long long>(d.cols, QSM_COLS), QSM_ELEMS / d.rows) : 0;
    if (t.strip || t.small) t.flat = 0;
    t.fchunks = t.flat ? (int)((d.rows * d.cols + QFLAT - 1) / QFLAT) : 0;
    t.strips = (int)((d.cols + QW - 1) / QW);
    const long long rcs = (d.rows + QR - 1) / QR;
    t.chunk0 = (int)chunk0;
    chunk0 += (d.rows > 0 && d.cols > 0) ? t.strips * rcs : 0;
    if (chunk0 > 0x7fffffffLL) return PS_EUNSUPPORTED;
    total_cols += psh::align_up((size_t)d.cols, 4);
  }
  return chunk0;
}

// Per-chunk tensor index: first the flat chunks (contiguous float4-addressable tensors), then the
// 64 x 256 tile chunks of the other tensors; chunk0 / fchunk0 of every tensor are set to its first
// chunk inside its own list.
static void build_chunk_maps(std::vector<QTensor>& ht, std::vector<int>& map, long long& fch,
                             long long& tch, long long* sch_out = nullptr, long long* nsm_out = nullptr) {
  fch = 0; tch = 0;
  for (size_t i = 0; i < ht.size(); ++i) {
    QTensor& t = ht[i];
    if (t.flat) { t.fchunk0 = (int)fch; fch += t.fchunks; }
  }
  map.reserve((size_t)fch);
  for (size_t i = 0; i < ht.size(); ++i)
    if (ht[i].flat) map.insert(map.end(), (size_t)ht[i].fchunks, (int)i);
  for (size_t i = 0; i < ht.size(); ++i) {
    QTensor& t = ht[i];
    if (t.flat || t.strip || t.small || t.rows <= 0 || t.cols <= 0) continue;
    const long long n = (long long)t.strips * ((t.rows + QR - 1) / QR);
    t.chunk0 = (int)tch;
    tch += n;
    map.insert(map.end(), (size_t)n, (int)i);
  }
  long long sch = 0;   // the register-resident strips (quantize only) come last in the table
  for (size_t i = 0; i < ht.size(); ++i) {
    QTensor& t = ht[i];
    if (!t.strip) continue;
    t.schunk0 = (int)sch;
    sch += t.schunks;
    map.insert(map.end(), (size_t)t.schunks, (int)i);
  }
  if (sch_out) *sch_out = sch;
  // the small tensors (quantize only): tensor indices, last in the table
  long long nsm = 0;
  for (size_t i = 0; i < ht.size(); ++i) {
    if (!ht[i].small) continue;
    for (long long c0 = 0; c0 < ht[i].cols; c0 += ht[i].small_w) {   // (tensor, first column) pairs
      map.push_back((int)i); map.push_back((int)c0); ++nsm;
    }
  }
  if (nsm_out) *nsm_out = nsm;
}

// Tensors without chunks would break the chunk -> tensor search (equal chunk0 keys resolve to
// the LAST tensor with that key, which is the one that owns the chunk, because empty tensors
// add no chunks); nothing else to do for them.

}  // namespace psk

using namespace psk;

// upper bound of the per-chunk table (a tensor is in one of the two chunk lists)
static size_t quant_map_bytes(const ps_quant_desc* desc, int count) {
  size_t n = 0;
  for (int i = 0; i < count; ++i) {
    if (desc[i].rows <= 0 || desc[i].cols <= 0) continue;
    const size_t flat = (size_t)((desc[i].rows * desc[i].cols + QFLAT - 1) / QFLAT);
    const size_t tile = (size_t)((desc[i].cols + QW - 1) / QW) * (size_t)((desc[i].rows + QR - 1) / QR);
    const size_t strips = 4 * (size_t)((desc[i].cols + QS_COLS - 1) / QS_COLS);
    // (small tensors: one pair of words per range of >= 64 columns, or one pair in all)
    n += std::max(std::max(flat, tile), strips) + 2 * ((size_t)desc[i].cols / 64 + 1);
  }
  return psh::align_up(sizeof(int) * (n + 1), 256);
}

static int quant_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
        prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
  }
  return cus;
}

extern "C" size_t ps_quantize_workspace_bytes(const ps_quant_desc* desc, int count) {
  if (!desc || count <= 0) return 0;
  size_t cols = 0;
  // per tensor: column maxima + one arrival counter per strip of 64 columns (tall strips); + the strip queue's counter
  for (int i = 0; i < count; ++i) {
    const size_t c = (size_t)(desc[i].cols > 0 ? desc[i].cols : 0);
    cols += psh::align_up(c, 4) + (c + QS_COLS - 1) / QS_COLS;
  }
  return psh::align_up(sizeof(QTensor) * count, 256) + psh::align_up(sizeof(unsigned) * (cols + 4), 256) +
         quant_map_bytes(desc, count) + 1024;
}

extern "C" int ps_quantize_f32(void* stream, const ps_quant_desc* desc, int count, void* workspace,
                               size_t workspace_bytes) {
  PS_DEVICE_CHECK();
  if (!desc || count <= 0 || !workspace) return PS_EINVAL;
  if (workspace_bytes < ps_quantize_workspace_bytes(desc, count)) return PS_EWORKSPACE;
  std::vector<QTensor> ht;
  size_t total_cols = 0;
  const long long chunks = build_tensors(desc, count, true, ht, total_cols);
  if (chunks < 0) return (int)chunks;
  if (chunks == 0) return PS_OK;
  hipStream_t st = (hipStream_t)stream;
  psh::Arena ar(workspace, workspace_bytes);
  QTensor* dt = ar.take<QTensor>(count);
  size_t team_strips = 0;
  for (int i = 0; i < count; ++i)
    if (ht[i].team > 1) team_strips += (size_t)ht[i].schunks / (size_t)ht[i].team;
  // column maxima | the strip queue's counter (4 words) | arrival counters of the tall strips: one memset
  unsigned* colmax = ar.take<unsigned>(total_cols + 4 + team_strips);
  if (ar.overflow) return PS_EWORKSPACE;
  size_t off = 0, soff = 0;
  for (int i = 0; i < count; ++i) {
    ht[i].colmax = colmax + off;
    off += psh::align_up((size_t)ht[i].cols, 4);
    if (ht[i].team > 1) {
      ht[i].sarrive = colmax + total_cols + 4 + soff;
      soff += (size_t)ht[i].schunks / (size_t)ht[i].team;
    }
  }
  PS_HIP(hipMemsetAsync(colmax, 0, sizeof(unsigned) * (total_cols + 4 + team_strips), st));
  // flat tensors: one launch per pass over chunks of consecutive elements; the others (odd sizes,
  // strided views) keep the tile kernels
  std::vector<int> hmap;
  long long fch = 0, tch = 0, sch = 0, nsm = 0;
  build_chunk_maps(ht, hmap, fch, tch, &sch, &nsm);
  int* dmap = ar.take<int>(hmap.size() + 1);
  if (ar.overflow) return PS_EWORKSPACE;
  PS_RC(psh::upload_async(st, dt, ht.data(), sizeof(QTensor) * count));
  PS_RC(psh::upload_async(st, dmap, hmap.data(), sizeof(int) * hmap.size()));
  const dim3 blk(256);
  if (sch + nsm > 0) {
    // the small tensors' workgroups, then the persistent ones: one per CU, the strips beyond the first 256 from a
    // queue (the counter behind the column maxima)
    const long long grid = nsm + std::min<long long>(sch, quant_cus());
    if (grid > 0x7fffffffLL) return PS_EUNSUPPORTED;
    hipLaunchKernelGGL(quant_strip_kernel, dim3((unsigned)grid), dim3(QS_THREADS), 0, st, dt, dmap + fch + tch,
                       (int)sch, colmax + total_cols, dmap + fch + tch + sch, (int)nsm);
  }
  if (fch > 0) {
    hipLaunchKernelGGL(quant_flat_kernel<0>, dim3((unsigned)fch), blk, 0, st, dt, dmap);
    hipLaunchKernelGGL(quant_flat_kernel<1>, dim3((unsigned)fch), blk, 0, st, dt, dmap);
  }
  if (tch > 0) {
    hipLaunchKernelGGL(quant_colmax_kernel, dim3((unsigned)tch), blk, 0, st, dt, dmap + fch, 0);
    hipLaunchKernelGGL(quant_encode_kernel, dim3((unsigned)tch), blk, 0, st, dt, dmap + fch, 0);
  }
  PS_LAUNCH_CHECK();
  return PS_OK;
}

extern "C" size_t ps_dequantize_workspace_bytes(const ps_quant_desc* desc, int count) {
  if (!desc || count <= 0) return 0;
  return psh::align_up(sizeof(QTensor) * count, 256) + quant_map_bytes(desc, count) + 1024;
}

extern "C" int ps_dequantize_f32(void* stream, const ps_quant_desc* desc, int count, void* workspace,
                                 size_t workspace_bytes) {
  PS_DEVICE_CHECK();
  if (!desc || count <= 0 || !workspace) return PS_EINVAL;
  if (workspace_bytes < ps_dequantize_workspace_bytes(desc, count)) return PS_EWORKSPACE;
  std::vector<QTensor> ht;
  size_t total_cols = 0;
  const long long chunks = build_tensors(desc, count, false, ht, total_cols);
  if (chunks < 0) return (int)chunks;
  if (chunks == 0) return PS_OK;
  hipStream_t st = (hipStream_t)stream;
  psh::Arena ar(workspace, workspace_bytes);
  QTensor* dt = ar.take<QTensor>(count);
  if (ar.overflow) return PS_EWORKSPACE;
  std::vector<int> hmap;
  long long fch = 0, tch = 0;
  build_chunk_maps(ht, hmap, fch, tch);
  int* dmap = ar.take<int>(hmap.size() + 1);
  if (ar.overflow) return PS_EWORKSPACE;
  PS_RC(psh::upload_async(st, dt, ht.data(), sizeof(QTensor) * count));
  PS_RC(psh::upload_async(st, dmap, hmap.data(), sizeof(int) * hmap.size()));
  if (fch + tch > 0x7fffffffLL) return PS_EUNSUPPORTED;
  hipLaunchKernelGGL(quant_decode_all_kernel, dim3((unsigned)(fch + tch)), dim3(256), 0, st, dt, dmap, (int)fch,
                     (int)tch);
  PS_LAUNCH_CHECK();
  return PS_OK;
}
