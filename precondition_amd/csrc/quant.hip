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
};

__device__ inline const QTensor* find_tensor(const QTensor* ts, int count, int chunk) {
  int lo = 0, hi = count - 1;
  while (lo < hi) {  // last tensor with chunk0 <= chunk
    const int mid = (lo + hi + 1) >> 1;
    if (ts[mid].chunk0 <= chunk) lo = mid; else hi = mid - 1;
  }
  return &ts[lo];
}

typedef float qf4 __attribute__((ext_vector_type(4)));
struct F4 { float x, y, z, w; };
__device__ inline F4 ldg4(const float* p) {
  const qf4 v = *(const qf4 PS_GLOBAL*)(p);
  return {v[0], v[1], v[2], v[3]};
}
__device__ inline float ldg1(const float* p) { return *(const float PS_GLOBAL*)(p); }

// ---- pass 1: column maxima -------------------------------------------------------------
__global__ __launch_bounds__(256) void quant_colmax_kernel(const QTensor* ts, int count) {
  __shared__ unsigned red[4][QW];
  const QTensor* t = find_tensor(ts, count, blockIdx.x);
  const int local = blockIdx.x - t->chunk0;
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
      // four rows in flight per lane (independent 16-byte loads) before the first use
      for (long long rb = r0 + wave; rb < r1; rb += 16) {
        F4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const long long r = rb + 4 * u;
          v[u] = r < r1 ? ldg4(t->fin + r * ld + col) : F4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const long long r = rb + 4 * u;
          float x[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (t->extract && col + j == r) x[j] = 0.f;
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
    const long long c = (long long)strip * QW + tid;
    if (c >= cols) return;
    unsigned m = 0u;
    for (long long r = r0; r < r1; ++r) {
      float x = ldg1(t->fin + r * ld + c);
      if (t->extract && c == r) x = 0.f;
      const unsigned b = __float_as_uint(fabsf(x));
      m = b > m ? b : m;
    }
    if (m) atomicMax(t->colmax + c, m);
  }
}

__device__ inline int encode1(float x, float bs_nonzero) {
  return (int)rintf(__fdiv_rn(x, bs_nonzero));  // QU:91-94; rint = round half to even
}

// ---- pass 2: bucket sizes, diagonal, codes ---------------------------------------------
__global__ __launch_bounds__(256) void quant_encode_kernel(const QTensor* ts, int count) {
  const QTensor* t = find_tensor(ts, count, blockIdx.x);
  const int local = blockIdx.x - t->chunk0;
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
    for (long long rb = r0 + wave; rb < r1; rb += 16) {
      F4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long r = rb + 4 * u;
        v[u] = r < r1 ? ldg4(t->fin + r * ld + col) : F4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long r = rb + 4 * u;
        if (r >= r1) break;
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
    if (rc == 0) t->bucket[c] = bs;
    for (long long r = r0; r < r1; ++r) {
      float x = ldg1(t->fin + r * ld + c);
      if (t->extract && c == r) { t->diag[r] = x; x = __fsub_rn(x, x); }
      const int q = encode1(x, bnz);
      if (t->bits == 16) reinterpret_cast<short*>(t->codes)[r * ldq + c] = (short)q;
      else reinterpret_cast<signed char*>(t->codes)[r * ldq + c] = (signed char)q;
    }
  }
}

// ---- to_float --------------------------------------------------------------------------
__global__ __launch_bounds__(256) void quant_decode_kernel(const QTensor* ts, int count) {
  const QTensor* t = find_tensor(ts, count, blockIdx.x);
  const int local = blockIdx.x - t->chunk0;
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
    for (long long r = r0 + wave; r < r1; r += 4) {
      int q[4];
      if (t->bits == 16) {
        const uint2 pk = *reinterpret_cast<const uint2*>(reinterpret_cast<const short*>(t->codes) + r * ldq + col);
        q[0] = (short)(pk.x & 0xffffu); q[1] = (short)(pk.x >> 16);
        q[2] = (short)(pk.y & 0xffffu); q[3] = (short)(pk.y >> 16);
      } else {
        const unsigned pk = *reinterpret_cast<const unsigned*>(reinterpret_cast<const signed char*>(t->codes) + r * ldq + col);
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
  } else {
    const long long c = (long long)strip * QW + tid;
    if (c >= cols) return;
    const float bs = t->bucket[c];
    for (long long r = r0; r < r1; ++r) {
      const int q = t->bits == 16 ? (int)reinterpret_cast<const short*>(t->codes)[r * ldq + c]
                                  : (int)reinterpret_cast<const signed char*>(t->codes)[r * ldq + c];
      float x = __fmul_rn((float)q, bs);
      if (t->extract && c == r) x = __fadd_rn(x, t->diag[r]);
      t->fout[r * ld + c] = x;
    }
  }
}

static bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// Validates descriptors and fills the device table image.  Returns total chunks or <0.
static long long build_tensors(const ps_quant_desc* desc, int count, bool encode,
                               std::vector<QTensor>& ht, size_t& total_cols) {
  ht.resize(count);
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
    t.strips = (int)((d.cols + QW - 1) / QW);
    const long long rcs = (d.rows + QR - 1) / QR;
    t.chunk0 = (int)chunk0;
    chunk0 += (d.rows > 0 && d.cols > 0) ? t.strips * rcs : 0;
    if (chunk0 > 0x7fffffffLL) return PS_EUNSUPPORTED;
    total_cols += psh::align_up((size_t)d.cols, 4);
  }
  (void)encode;
  return chunk0;
}

// Tensors without chunks would break the chunk -> tensor search (equal chunk0 keys resolve to
// the LAST tensor with that key, which is the one that owns the chunk, because empty tensors
// add no chunks); nothing else to do for them.

}  // namespace psk

using namespace psk;

extern "C" size_t ps_quantize_workspace_bytes(const ps_quant_desc* desc, int count) {
  if (!desc || count <= 0) return 0;
  size_t cols = 0;
  for (int i = 0; i < count; ++i) cols += psh::align_up((size_t)(desc[i].cols > 0 ? desc[i].cols : 0), 4);
  return psh::align_up(sizeof(QTensor) * count, 256) + psh::align_up(sizeof(unsigned) * cols, 256) + 1024;
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
  unsigned* colmax = ar.take<unsigned>(total_cols);
  if (ar.overflow) return PS_EWORKSPACE;
  size_t off = 0;
  for (int i = 0; i < count; ++i) {
    ht[i].colmax = colmax + off;
    off += psh::align_up((size_t)ht[i].cols, 4);
  }
  PS_HIP(hipMemsetAsync(colmax, 0, sizeof(unsigned) * total_cols, st));
  PS_RC(psh::upload_async(st, dt, ht.data(), sizeof(QTensor) * count));
  const dim3 grid((unsigned)chunks), blk(256);
  hipLaunchKernelGGL(quant_colmax_kernel, grid, blk, 0, st, dt, count);
  hipLaunchKernelGGL(quant_encode_kernel, grid, blk, 0, st, dt, count);
  PS_LAUNCH_CHECK();
  return PS_OK;
}

extern "C" size_t ps_dequantize_workspace_bytes(const ps_quant_desc* desc, int count) {
  if (!desc || count <= 0) return 0;
  return psh::align_up(sizeof(QTensor) * count, 256) + 1024;
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
  PS_RC(psh::upload_async(st, dt, ht.data(), sizeof(QTensor) * count));
  hipLaunchKernelGGL(quant_decode_kernel, dim3((unsigned)chunks), dim3(256), 0, st, dt, count);
  PS_LAUNCH_CHECK();
  return PS_OK;
}
