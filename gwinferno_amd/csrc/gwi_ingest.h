// gwi_ingest.h -- the setup path on the device (include/gwi_engine.h "setup on the device").
//
// The reference prepares its theta-independent per-sample quantities with NumPy/JAX when the model objects are built
// (masks: models/bsplines/single.py:54-55, distributions.py:119,143,162, parametric.py:141-145; logarithms:
// interpolation.py:357,447; dVc/dz by table interpolation: cosmology.py:95-120, parametric.py:116; 1/prior:
// examples/simple_bspline_example.py:58-71).  Here ONE kernel evaluates a small straight-line register program for every
// sample of a set: raw catalog columns in (fp64 or fp32), the engine's columns -- kappa included -- out, each sample
// read once and each column written once (coalesced: lane = sample).  The program is the same for all lanes, so its
// decoding runs on the scalar unit; the registers of a lane live in a small private array (the one kernel of this
// library that uses scratch: it runs once per catalog, not once per evaluation).
//
// Bit-exactness against the host evaluation of the same program (gwinferno_amd/expr.py): contraction is OFF in this
// file's device code, comparisons / selections / sqrt / division are IEEE, the table interpolation follows
// numpy.interp's operation order; LOG and LOG1P come from the device maths library (<= 1 ulp).
#pragma once

#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <vector>

#include "gwi_engine.h"

namespace gwi {

struct IngestArgs {
  const gwi_ingest_op* ops;
  long long n, stride;  // stride = threads of the launch (the kernel reads no implicit argument)
  int n_ops, n_regs;
  const void* src[GWI_INGEST_MAX_SOURCES];
  const double* tab[GWI_INGEST_MAX_TABLES];
  double* out[GWI_MAX_COLS];
  int tab_len[GWI_INGEST_MAX_TABLES];
  int dtype[GWI_INGEST_MAX_SOURCES];
};

// numpy.interp (numpy/_core/src/multiarray/compiled_base.c, arr_interp) with its default end values
__device__ inline double ing_interp(double x, const double* __restrict__ xp, const double* __restrict__ fp, int n) {
#pragma clang fp contract(off)
  if (x != x) return x;
  if (x > xp[n - 1]) return fp[n - 1];
  if (x < xp[0]) return fp[0];
  int lo = 0, hi = n - 1;  // xp[lo] <= x <= xp[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (xp[mid] <= x) lo = mid; else hi = mid;
  }
  const int j = (xp[hi] <= x) ? hi : lo;  // largest j with xp[j] <= x
  if (j == n - 1) return fp[j];
  if (xp[j] == x) return fp[j];
  const double slope = (fp[j + 1] - fp[j]) / (xp[j + 1] - xp[j]);
  double r = slope * (x - xp[j]) + fp[j];
  if (r != r) {
    r = slope * (x - xp[j + 1]) + fp[j + 1];
    if (r != r && fp[j] == fp[j + 1]) r = fp[j];
  }
  return r;
}

// j + f: the piece and weight numpy.interp uses (gwinferno_amd/expr.py "gridindex"); NaN stays NaN
__device__ inline double ing_gridindex(double x, const double* __restrict__ g, int n) {
#pragma clang fp contract(off)
  int cnt;  // numpy.searchsorted(g, x, side="right"): NaN sorts behind everything
  if (x != x) {
    cnt = n;
  } else {
    int lo = 0, hi = n;  // first index with g[idx] > x
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (g[mid] <= x) lo = mid + 1; else hi = mid;
    }
    cnt = lo;
  }
  int j = cnt - 1;
  j = j < 0 ? 0 : (j > n - 2 ? n - 2 : j);
  const double v = (x - g[j]) / (g[j + 1] - g[j]);
  const double f = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
  return (double)j + f;
}

__global__ __launch_bounds__(256) void ingest_kernel(const IngestArgs a) {
#pragma clang fp contract(off)
  double r[GWI_INGEST_MAX_REGS];
  for (int k = 0; k < a.n_regs; ++k) r[k] = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.n; i += a.stride) {
    for (int o = 0; o < a.n_ops; ++o) {
      const gwi_ingest_op op = a.ops[o];  // wave-uniform: scalar loads
      const double x = r[op.a & (GWI_INGEST_MAX_REGS - 1)], y = r[op.b & (GWI_INGEST_MAX_REGS - 1)], z = r[op.c & (GWI_INGEST_MAX_REGS - 1)];
      double v = 0.0;
      switch (op.op) {
        case GWI_ING_LOAD:
          v = a.dtype[op.a] == GWI_DTYPE_F32 ? (double)static_cast<const float*>(a.src[op.a])[i] : static_cast<const double*>(a.src[op.a])[i];
          break;
        case GWI_ING_CONST: v = op.k; break;
        case GWI_ING_LOG: v = log(x); break;
        case GWI_ING_LOG1P: v = log1p(x); break;
        case GWI_ING_NEG: v = -x; break;
        case GWI_ING_ABS: v = fabs(x); break;
        case GWI_ING_NOT: v = x != 0.0 ? 0.0 : 1.0; break;
        case GWI_ING_SQRT: v = __builtin_sqrt(x); break;
        case GWI_ING_ISFINITE: v = (fabs(x) < __builtin_inf()) ? 1.0 : 0.0; break;  // false for NaN
        case GWI_ING_ADD: v = x + y; break;
        case GWI_ING_SUB: v = x - y; break;
        case GWI_ING_MUL: v = x * y; break;
        case GWI_ING_DIV: v = x / y; break;
        case GWI_ING_LT: v = x < y ? 1.0 : 0.0; break;
        case GWI_ING_GT: v = x > y ? 1.0 : 0.0; break;
        case GWI_ING_LE: v = x <= y ? 1.0 : 0.0; break;
        case GWI_ING_GE: v = x >= y ? 1.0 : 0.0; break;
        case GWI_ING_AND: v = (x != 0.0 && y != 0.0) ? 1.0 : 0.0; break;
        case GWI_ING_OR: v = (x != 0.0 || y != 0.0) ? 1.0 : 0.0; break;
        case GWI_ING_WHERE: v = x != 0.0 ? y : z; break;
        case GWI_ING_INTERP: v = ing_interp(x, a.tab[op.b], a.tab[op.c], a.tab_len[op.b]); break;
        case GWI_ING_GRIDINDEX: v = ing_gridindex(x, a.tab[op.b], a.tab_len[op.b]); break;
        case GWI_ING_STORE: a.out[op.dst][i] = x; continue;
        default: break;
      }
      r[op.dst & (GWI_INGEST_MAX_REGS - 1)] = v;
    }
  }
}

// Spline coordinate x -> knot coordinate u = (x - lo) / dx, once per catalog (gwi_device.h: spline_locate_knot): the scan
// kernels read u.  `top` >= 0: clamp into [0, top] (exponentiated splines without the zero-outside flag: every live sample lies
// in the closed domain already, x = hi lands one ulp below the number of intervals; NaN cannot occur -- non-finite column
// entries were parked when the catalog was bound -- and would become 0).  Same arithmetic as the scan used to do per sample.
struct KnotArgs {
  const double* x;
  double* u;
  long long n, stride;
  double lo, inv_dx, top;
};
__global__ __launch_bounds__(256) void spline_knot_kernel(const KnotArgs a) {
#pragma clang fp contract(off)
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.n; i += a.stride) {
    double u = (a.x[i] - a.lo) * a.inv_dx;
    if (a.top >= 0.0) u = fmin(fmax(u, 0.0), a.top);
    a.u[i] = u;
  }
}
inline hipError_t spline_knot_run(const double* x, double* u, long long n, double lo, double inv_dx, double top, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  long long blocks = (n + 255) / 256;
  if (blocks > 256LL * 16) blocks = 256LL * 16;
  const KnotArgs a{x, u, n, blocks * 256, lo, inv_dx, top};
  hipLaunchKernelGGL(spline_knot_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a);
  return hipGetLastError();
}

// Host side: validate, upload sources / tables / ops, run, release.  `d_out[c]` are device arrays of n doubles the
// caller owns.  Returns GWI_OK or a status with `err` filled in.
inline gwi_status ingest_check(std::string& err, const gwi_ingest_program* p, int n_cols) {
  auto bad = [&](const std::string& m) {
    err = "ingest program: " + m;
    return GWI_ERR_INVALID;
  };
  if (!p || !p->ops || p->n_ops < 1) return bad("empty");
  if (p->n_regs < 0 || p->n_regs > GWI_INGEST_MAX_REGS) return bad("more than GWI_INGEST_MAX_REGS registers");
  if (p->n_sources < 0 || p->n_sources > GWI_INGEST_MAX_SOURCES) return bad("more than GWI_INGEST_MAX_SOURCES sources");
  if (p->n_tables < 0 || p->n_tables > GWI_INGEST_MAX_TABLES) return bad("more than GWI_INGEST_MAX_TABLES tables");
  if (n_cols < 1 || n_cols > GWI_MAX_COLS) return bad("column count out of range");
  if (p->n_sources && (!p->sources || !p->source_dtype)) return bad("null source table");
  if (p->n_tables && (!p->tables || !p->table_len)) return bad("null interpolation tables");
  for (int s = 0; s < p->n_sources; ++s)
    if (!p->sources[s] || (p->source_dtype[s] != GWI_DTYPE_F64 && p->source_dtype[s] != GWI_DTYPE_F32)) return bad("null source or unknown source dtype");
  for (int t = 0; t < p->n_tables; ++t)
    if (!p->tables[t] || p->table_len[t] < 2 || p->table_len[t] > (1LL << 30)) return bad("an interpolation table needs 2 .. 2^30 entries");
  std::vector<char> stored(n_cols, 0);
  // a register must have been written by an earlier op of the program before it is read: the kernel's registers carry a
  // lane's values from one sample of its grid-stride loop to the next, so a program that reads first would see the
  // previous sample's data, silently
  std::vector<char> written(GWI_INGEST_MAX_REGS, 0);
  bool read_unwritten = false;
  auto reg = [&](int v) { return v >= 0 && v < p->n_regs; };
  auto src = [&](int v) {  // a register operand that is read
    if (!reg(v)) return false;
    if (!written[v]) read_unwritten = true;
    return true;
  };
  for (int o = 0; o < p->n_ops; ++o) {
    const gwi_ingest_op& op = p->ops[o];
    const std::string at = "op " + std::to_string(o) + ": ";
    switch (op.op) {
      case GWI_ING_LOAD:
        if (!reg(op.dst) || op.a < 0 || op.a >= p->n_sources) return bad(at + "LOAD out of range");
        break;
      case GWI_ING_CONST:
        if (!reg(op.dst)) return bad(at + "register out of range");
        break;
      case GWI_ING_LOG: case GWI_ING_LOG1P: case GWI_ING_NEG: case GWI_ING_ABS: case GWI_ING_NOT: case GWI_ING_SQRT: case GWI_ING_ISFINITE:
        if (!reg(op.dst) || !src(op.a)) return bad(at + "register out of range");
        break;
      case GWI_ING_ADD: case GWI_ING_SUB: case GWI_ING_MUL: case GWI_ING_DIV: case GWI_ING_LT: case GWI_ING_GT: case GWI_ING_LE: case GWI_ING_GE:
      case GWI_ING_AND: case GWI_ING_OR:
        if (!reg(op.dst) || !src(op.a) || !src(op.b)) return bad(at + "register out of range");
        break;
      case GWI_ING_WHERE:
        if (!reg(op.dst) || !src(op.a) || !src(op.b) || !src(op.c)) return bad(at + "register out of range");
        break;
      case GWI_ING_INTERP:
        if (!reg(op.dst) || !src(op.a) || op.b < 0 || op.b >= p->n_tables || op.c < 0 || op.c >= p->n_tables || p->table_len[op.b] != p->table_len[op.c])
          return bad(at + "INTERP needs two tables of one length");
        break;
      case GWI_ING_GRIDINDEX:
        if (!reg(op.dst) || !src(op.a) || op.b < 0 || op.b >= p->n_tables) return bad(at + "GRIDINDEX table out of range");
        break;
      case GWI_ING_STORE:
        if (!src(op.a) || op.dst < 0 || op.dst >= n_cols) return bad(at + "STORE out of range");
        stored[op.dst] = 1;
        break;
      default:
        return bad(at + "unknown opcode " + std::to_string(op.op));
    }
    if (read_unwritten) return bad(at + "reads a register no earlier op has written");
    if (op.op != GWI_ING_STORE) written[op.dst] = 1;
  }
  for (int c = 0; c < n_cols; ++c)
    if (!stored[c]) return bad("column " + std::to_string(c) + " is never stored");
  return GWI_OK;
}

inline gwi_status ingest_run(std::string& err, const gwi_ingest_program* p, long long n, int n_cols, double* const* d_out, hipStream_t stream) {
  gwi_status st = ingest_check(err, p, n_cols);
  if (st != GWI_OK || n == 0) return st;
  std::vector<void*> owned;
  auto release = [&]() {
    for (void* q : owned) (void)hipFree(q);
  };
  auto hip_fail = [&](const char* what, hipError_t e) {
    err = std::string("ingest: ") + what + ": " + hipGetErrorString(e);
    release();
    return GWI_ERR_HIP;
  };
  IngestArgs a;
  std::memset(&a, 0, sizeof(a));
  a.n = n;
  a.n_ops = p->n_ops;
  a.n_regs = p->n_regs;
  hipError_t e;
  for (int s = 0; s < p->n_sources; ++s) {
    const size_t bytes = (size_t)n * (p->source_dtype[s] == GWI_DTYPE_F32 ? 4 : 8);
    void* d = nullptr;
    if ((e = hipMalloc(&d, bytes)) != hipSuccess) return hip_fail("hipMalloc(source)", e);
    owned.push_back(d);
    if ((e = hipMemcpyAsync(d, p->sources[s], bytes, hipMemcpyHostToDevice, stream)) != hipSuccess) return hip_fail("upload of a source column", e);
    a.src[s] = d;
    a.dtype[s] = p->source_dtype[s];
  }
  for (int t = 0; t < p->n_tables; ++t) {
    void* d = nullptr;
    const size_t bytes = sizeof(double) * (size_t)p->table_len[t];
    if ((e = hipMalloc(&d, bytes)) != hipSuccess) return hip_fail("hipMalloc(table)", e);
    owned.push_back(d);
    if ((e = hipMemcpyAsync(d, p->tables[t], bytes, hipMemcpyHostToDevice, stream)) != hipSuccess) return hip_fail("upload of a table", e);
    a.tab[t] = static_cast<const double*>(d);
    a.tab_len[t] = (int)p->table_len[t];
  }
  {
    void* d = nullptr;
    const size_t bytes = sizeof(gwi_ingest_op) * (size_t)p->n_ops;
    if ((e = hipMalloc(&d, bytes)) != hipSuccess) return hip_fail("hipMalloc(ops)", e);
    owned.push_back(d);
    if ((e = hipMemcpyAsync(d, p->ops, bytes, hipMemcpyHostToDevice, stream)) != hipSuccess) return hip_fail("upload of the program", e);
    a.ops = static_cast<const gwi_ingest_op*>(d);
  }
  for (int c = 0; c < n_cols; ++c) a.out[c] = d_out[c];
  long long blocks = (n + 255) / 256;
  if (blocks > 256LL * 16) blocks = 256LL * 16;  // 16 workgroups per CU, grid-stride beyond
  a.stride = blocks * 256;
  hipLaunchKernelGGL(ingest_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a);
  if ((e = hipGetLastError()) != hipSuccess) return hip_fail("launch of ingest_kernel", e);
  if ((e = hipStreamSynchronize(stream)) != hipSuccess) return hip_fail("ingest_kernel", e);
  release();
  return GWI_OK;
}

}  // namespace gwi
