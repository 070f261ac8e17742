// xsmm_generator.cpp -- the text-generator front door of the engine: sparse and dense kernels as HIP source for gfx950,
// MatrixMarket readers, libxsmm_strerror, and the executable form of the sparse text kernels (hiprtc).
//
// Reference: src/generator_spgemm.c (mux :55-238, file front door :246-450), src/generator_spgemm_csr_asparse.c:46-151,
// src/generator_spgemm_csc_bsparse.c:85-189, src/generator_spgemm_csc_asparse.c:223-349, the readers
// src/generator_spgemm_csr_reader.c:46-170 / src/generator_spgemm_csc_reader.c:85-215, src/generator_gemm.c:51-330 and
// src/generator_common.c (string buffer :66-123, signature :756-792, libxsmm_strerror).
//
// The reference emits, per sparsity pattern, a C function with one statement per non-zero (the pattern is code, the
// values of the sparse operand stay an argument). Here the same thing is emitted as a HIP kernel over a *batch*: one
// thread per (item, element of the vectorised dimension) keeps the dense operand's used entries and its slice of C in
// registers and walks the baked-in pattern; the sparse values are wave-uniform (scalar loads). Per C element the
// operations and their order are the reference's: c = c + a*b in pattern order (or fma, what a contracting compiler makes
// of that statement on an FMA machine -- selectable).
#include "xsmm_internal.hpp"

#include <cstdarg>
#include <cstring>
#include <string>
#include <vector>

using namespace xsmm;

namespace {

enum { // src/generator_common.h:267-320
  ERR_GENERAL = 90000, ERR_ALLOC = 90001, ERR_BUFFER_TOO_SMALL = 90002, ERR_APPEND_STR = 90003, ERR_ARCH_PREC = 90004, ERR_ARCH = 90005,
  ERR_UNSUP_ARCH = 90006, ERR_LDA = 90007, ERR_LDB = 90008, ERR_LDC = 90009, ERR_SPGEMM_GEN = 90010, ERR_CSC_INPUT = 90011,
  ERR_CSC_READ_LEN = 90012, ERR_CSC_READ_DESC = 90013, ERR_CSC_READ_ELEMS = 90014, ERR_CSC_LEN = 90015, ERR_CSC_ALLOC_DATA = 90033,
  ERR_CSR_ALLOC_DATA = 90034, ERR_CSR_INPUT = 90035, ERR_CSR_READ_LEN = 90036, ERR_CSR_READ_DESC = 90037, ERR_CSR_READ_ELEMS = 90038,
  ERR_CSR_LEN = 90039, ERR_UNSUP_DATATYPE = 90049, ERR_INVALID_GEMM_CONFIG = 90051, ERR_UNIQUE_VAL = 90052
};

void fail(libxsmm_generated_code* io, unsigned code) { if (nullptr != io) io->last_error = code; }

// string buffer semantics of libxsmm_append_code_as_string: malloc'ed, NUL terminated, replaced on every append
void append(libxsmm_generated_code* io, const std::string& text)
{
  if (nullptr == io) return;
  if (io->code_type > 1) { fail(io, ERR_APPEND_STR); return; }
  const size_t old = (nullptr != io->generated_code ? io->code_size : 0);
  char* const fresh = static_cast<char*>(malloc(old + text.size() + 1));
  if (nullptr == fresh) { fail(io, ERR_ALLOC); return; }
  if (0 < old) memcpy(fresh, io->generated_code, old);
  memcpy(fresh + old, text.data(), text.size());
  fresh[old + text.size()] = 0;
  if (0 < old) free(io->generated_code);
  io->generated_code = fresh;
  io->code_size = (unsigned)(old + text.size());
  io->buffer_size = io->code_size + 1;
}

std::string fmt(const char* f, ...)
{
  char buf[512];
  va_list ap; va_start(ap, f);
  vsnprintf(buf, sizeof(buf), f, ap);
  va_end(ap);
  return buf;
}

// ---- MatrixMarket coordinate files ------------------------------------------------------------------------------------
// '%' lines are comments; the first other line is "rows cols nnz"; then 1-based "row col value" triples which must come
// grouped by row (CSR) / column (CSC): ptr[major + 1] is simply the running count; majors without entries are
// back-filled afterwards.
unsigned read_coordinate_file(const char* path, bool csr, std::vector<unsigned>& ptr, std::vector<unsigned>& idx, std::vector<double>& values,
                              unsigned& rows, unsigned& cols, unsigned& nnz)
{
  FILE* const f = (nullptr != path ? fopen(path, "r") : nullptr);
  if (nullptr == f) return csr ? ERR_CSR_INPUT : ERR_CSC_INPUT;
  char line[513];
  bool header = false;
  unsigned count = 0, nmajor = 0, err = 0;
  std::vector<char> seen;
  while (0 == err && nullptr != fgets(line, 512, f)) {
    if (511 <= strlen(line)) { err = csr ? ERR_CSR_READ_LEN : ERR_CSC_READ_LEN; break; }
    if ('%' == line[0]) continue;
    if (!header) {
      if (3 != sscanf(line, "%u %u %u", &rows, &cols, &nnz) || 0 == rows || 0 == cols || 0 == nnz) { err = csr ? ERR_CSR_READ_DESC : ERR_CSC_READ_DESC; break; }
      nmajor = csr ? rows : cols;
      ptr.assign((size_t)nmajor + 1, nnz); ptr[0] = 0;
      idx.assign(nnz, 0); values.assign(nnz, 0.0); seen.assign(nmajor, 0);
      header = true;
    }
    else {
      unsigned r = 0, c = 0; double v = 0;
      if (3 != sscanf(line, "%u %u %lf", &r, &c, &v) || 0 == r || 0 == c || count >= nnz) { err = csr ? ERR_CSR_READ_ELEMS : ERR_CSC_READ_ELEMS; break; }
      const unsigned major = (csr ? r : c) - 1, minor = (csr ? c : r) - 1;
      if (major >= nmajor) { err = csr ? ERR_CSR_READ_ELEMS : ERR_CSC_READ_ELEMS; break; }
      idx[count] = minor; values[count] = v; ++count;
      seen[major] = 1; ptr[major + 1] = count;
    }
  }
  fclose(f);
  if (0 == err && (!header || count != nnz)) err = csr ? ERR_CSR_LEN : ERR_CSC_LEN;
  if (0 != err) return err;
  for (unsigned i = 0; i < nmajor; ++i) if (0 == seen[i]) ptr[i + 1] = ptr[i];
  return 0;
}

// ---- kernel text -------------------------------------------------------------------------------------------------------
enum SpKind { SP_CSR_ASPARSE = 0, SP_CSC_BSPARSE = 1, SP_CSC_ASPARSE = 2,
              SP_SOA_ASPARSE = 3, SP_SOA_BSPARSE = 4, SP_SOA_RM_AC = 5, SP_SOA_RM_BC = 6 }; // SOA: [row][col][v] operands

struct SpShape { int typesize, m, n, k, lda, ldb, ldc, beta0, v; }; // v: SOA width (0: plain matrices)

const char* tname(int typesize) { return 8 == typesize ? "double" : "float"; }

// threads per item
int sp_lanes(SpKind kind, const SpShape& s)
{
  switch (kind) {
    case SP_CSR_ASPARSE: return (0 != s.beta0 ? s.ldc : s.n); // beta == 0 clears ldc (not n) entries per row (:79)
    case SP_CSC_BSPARSE: return s.m;
    case SP_CSC_ASPARSE: return s.n;
    case SP_SOA_ASPARSE: case SP_SOA_RM_BC: return s.n * s.v; // one thread per (column, run)
    default: return s.m * s.v;                                 // SP_SOA_BSPARSE, SP_SOA_RM_AC: one thread per (row, run)
  }
}

// ---- SOA kernels (EDGE/SeisSol fused runs): element-wise in the innermost index v ---------------------------------------
// reference: src/generator_spgemm_csr_asparse_soa.c:212-330 (rows of A without non-zeros are left alone),
// src/generator_spgemm_csc_bsparse_soa.c:177-420 and src/generator_spgemm_csr_bsparse_soa.c:160-330 (every C column is
// loaded or zeroed and stored; per k the first pattern entry (k, n) contributes), src/generator_gemm_rm_{ac,bc}_soa.c.
// All of them: register accumulator seeded with C (or 0), fused multiply-add, k ascending.
struct SoaEntry { unsigned k, p; }; // contribution of B(k, n): p indexes the shared operand (values array / plain matrix)

std::string soa_body(SpKind kind, const SpShape& s, const unsigned* ptr, const unsigned* idx, bool csr)
{
  std::string t;
  const int lanes = sp_lanes(kind, s), V = s.v;
  t += "  const long long xs_gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;\n";
  t += fmt("  const long long xs_item = xs_gid / %d;\n  const int xs_l = (int)(xs_gid - xs_item * %d);\n", lanes, lanes);
  t += "  if (xs_item >= batch) return;\n";
  if (SP_SOA_ASPARSE == kind || SP_SOA_RM_BC == kind) { // C[m][n][v] (+)= A(m,k) * B[k][n][v]; thread = (n, v)
    t += "  const T* const b = B + xs_item * stride_dense + xs_l;\n  T* const c = C + xs_item * stride_c + xs_l;\n";
    std::vector<char> used(s.k > 0 ? s.k : 1, 0);
    if (SP_SOA_RM_BC == kind) used.assign(used.size(), 1);
    else for (int m = 0; m < s.m; ++m) for (unsigned p = ptr[m]; p < ptr[m + 1]; ++p) used[idx[p]] = 1;
    for (int k = 0; k < s.k; ++k) if (used[k]) t += fmt("  const T b%d = b[%d];\n", k, k * s.ldb * V);
    for (int m = 0; m < s.m; ++m) {
      if (SP_SOA_ASPARSE == kind && ptr[m] == ptr[m + 1]) continue; // untouched, also for beta == 0
      t += fmt("  { T acc = %s;\n", 0 != s.beta0 ? "(T)0" : fmt("c[%d]", m * s.ldc * V).c_str());
      if (SP_SOA_ASPARSE == kind) for (unsigned p = ptr[m]; p < ptr[m + 1]; ++p) t += fmt("    acc = XACC(A[%u], b%u, acc);\n", p, idx[p]);
      else for (int k = 0; k < s.k; ++k) t += fmt("    acc = XACC(A[%d], b%d, acc);\n", m * s.lda + k, k);
      t += fmt("    c[%d] = acc; }\n", m * s.ldc * V);
    }
  }
  else { // C[m][n][v] (+)= A[m][k][v] * B(k,n); thread = (m, v)
    t += fmt("  const int xs_m = xs_l / %d, xs_v = xs_l - xs_m * %d;\n", V, V);
    t += fmt("  const T* const a = A + xs_item * stride_dense + (long long)xs_m * %d + xs_v;\n", s.lda * V);
    t += fmt("  T* const c = C + xs_item * stride_c + (long long)xs_m * %d + xs_v;\n", s.ldc * V);
    std::vector<std::vector<SoaEntry>> cols(s.n > 0 ? s.n : 1);
    if (SP_SOA_RM_AC == kind) {
      for (int n = 0; n < s.n; ++n) for (int k = 0; k < s.k; ++k) cols[n].push_back(SoaEntry{ (unsigned)k, (unsigned)(k * s.ldb + n) });
    }
    else if (csr) { // rows of B are k: entries arrive k-ascending per column by construction
      for (int k = 0; k < s.k; ++k) for (unsigned p = ptr[k]; p < ptr[k + 1]; ++p) {
        const unsigned n = idx[p];
        if (n < (unsigned)s.n && (cols[n].empty() || cols[n].back().k != (unsigned)k)) cols[n].push_back(SoaEntry{ (unsigned)k, p });
      }
    }
    else { // columns of B: per k the first entry with that row index
      for (int n = 0; n < s.n; ++n) for (int k = 0; k < s.k; ++k) {
        for (unsigned p = ptr[n]; p < ptr[n + 1]; ++p) if (idx[p] == (unsigned)k) { cols[n].push_back(SoaEntry{ (unsigned)k, p }); break; }
      }
    }
    std::vector<char> used(s.k > 0 ? s.k : 1, 0);
    for (int n = 0; n < s.n; ++n) for (const SoaEntry& e : cols[n]) used[e.k] = 1;
    for (int k = 0; k < s.k; ++k) if (used[k]) t += fmt("  const T a%d = a[%d];\n", k, k * V);
    for (int n = 0; n < s.n; ++n) {
      t += fmt("  { T acc = %s;\n", 0 != s.beta0 ? "(T)0" : fmt("c[%d]", n * V).c_str());
      for (const SoaEntry& e : cols[n]) t += fmt("    acc = XACC(a%u, B[%u], acc);\n", e.k, e.p);
      t += fmt("    c[%d] = acc; }\n", n * V);
    }
  }
  return t;
}

// The statements between the signature and the closing brace. A, B, C, stride_dense, stride_c and batch are the kernel's
// arguments (sparse operand: the values array in storage order).
std::string spgemm_body(SpKind kind, const SpShape& s, const unsigned* ptr, const unsigned* idx)
{
  std::string t;
  const int lanes = sp_lanes(kind, s);
  t += "  const long long xs_gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;\n";
  t += fmt("  const long long xs_item = xs_gid / %d;\n  const int xs_l = (int)(xs_gid - xs_item * %d);\n", lanes, lanes);
  t += "  if (xs_item >= batch) return;\n";
  if (SP_CSR_ASPARSE == kind) { // row-major: C[m*ldc+n] += A[p] * B[col[p]*ldb+n]
    t += "  const T* const b = B + xs_item * stride_dense;\n  T* const c = C + xs_item * stride_c;\n  const int n = xs_l;\n";
    if (0 != s.beta0 && s.ldc > s.n) {
      t += fmt("  if (n >= %d) { /* beta == 0 clears the padding columns as well */\n", s.n);
      t += fmt("    for (int m = 0; m < %d; ++m) c[m * %d + n] = (T)0;\n    return;\n  }\n", s.m, s.ldc);
    }
    std::vector<char> used(s.k > 0 ? s.k : 1, 0);
    for (int m = 0; m < s.m; ++m) for (unsigned p = ptr[m]; p < ptr[m + 1]; ++p) if (idx[p] < (unsigned)s.k) used[idx[p]] = 1;
    for (int k = 0; k < s.k; ++k) if (used[k]) t += fmt("  const T b%d = b[%d + n];\n", k, k * s.ldb);
    for (int m = 0; m < s.m; ++m) {
      bool any = false;
      for (unsigned p = ptr[m]; p < ptr[m + 1]; ++p) any = any || (idx[p] < (unsigned)s.k);
      if (!any && 0 == s.beta0) continue;
      t += fmt("  { T acc = %s;\n", 0 != s.beta0 ? "(T)0" : fmt("c[%d + n]", m * s.ldc).c_str());
      for (unsigned p = ptr[m]; p < ptr[m + 1]; ++p) if (idx[p] < (unsigned)s.k) t += fmt("    acc = XACC(A[%u], b%u, acc);\n", p, idx[p]);
      t += fmt("    c[%d + n] = acc; }\n", m * s.ldc);
    }
  }
  else if (SP_CSC_BSPARSE == kind) { // column-major: C[n*ldc+m] += A[row[p]*lda+m] * B[p]
    t += "  const T* const a = A + xs_item * stride_dense;\n  T* const c = C + xs_item * stride_c;\n  const int m = xs_l;\n";
    std::vector<char> used(s.k > 0 ? s.k : 1, 0);
    for (int n = 0; n < s.n; ++n) for (unsigned p = ptr[n]; p < ptr[n + 1]; ++p) if (idx[p] < (unsigned)s.k) used[idx[p]] = 1;
    for (int k = 0; k < s.k; ++k) if (used[k]) t += fmt("  const T a%d = a[%d + m];\n", k, k * s.lda);
    for (int n = 0; n < s.n; ++n) {
      bool any = false;
      for (unsigned p = ptr[n]; p < ptr[n + 1]; ++p) any = any || (idx[p] < (unsigned)s.k);
      if (!any && 0 == s.beta0) continue;
      t += fmt("  { T acc = %s;\n", 0 != s.beta0 ? "(T)0" : fmt("c[%d + m]", n * s.ldc).c_str());
      for (unsigned p = ptr[n]; p < ptr[n + 1]; ++p) if (idx[p] < (unsigned)s.k) t += fmt("    acc = XACC(a%u, B[%u], acc);\n", idx[p], p);
      t += fmt("    c[%d + m] = acc; }\n", n * s.ldc);
    }
  }
  else { // column-major, A sparse by columns: C[n*ldc+row[p]] += A[p] * B[n*ldb+k]
    t += fmt("  const T* const b = B + xs_item * stride_dense + (long long)xs_l * %d;\n  T* const c = C + xs_item * stride_c + (long long)xs_l * %d;\n", s.ldb, s.ldc);
    std::vector<char> touched(s.m > 0 ? s.m : 1, 0);
    for (int k = 0; k < s.k; ++k) for (unsigned p = ptr[k]; p < ptr[k + 1]; ++p) if (idx[p] < (unsigned)s.m) touched[idx[p]] = 1;
    for (int m = 0; m < s.m; ++m) {
      if (0 != s.beta0) t += fmt("  T c%d = (T)0;\n", m);
      else if (touched[m]) t += fmt("  T c%d = c[%d];\n", m, m);
    }
    for (int k = 0; k < s.k; ++k) {
      bool any = false;
      for (unsigned p = ptr[k]; p < ptr[k + 1]; ++p) any = any || (idx[p] < (unsigned)s.m);
      if (!any) continue;
      t += fmt("  { const T bk = b[%d];\n", k);
      for (unsigned p = ptr[k]; p < ptr[k + 1]; ++p) if (idx[p] < (unsigned)s.m) t += fmt("    c%u = XACC(A[%u], bk, c%u);\n", idx[p], p, idx[p]);
      t += "  }\n";
    }
    for (int m = 0; m < s.m; ++m) if (0 != s.beta0 || touched[m]) t += fmt("  c[%d] = c%d;\n", m, m);
  }
  return t;
}

std::string spgemm_prologue(int typesize, int fma)
{
  std::string t = "// generated by libxsmm-amd (sparse text kernel for gfx950; pattern baked in, values are an argument)\n";
  t += std::string("typedef ") + tname(typesize) + " T;\n";
  if (0 != fma) t += (8 == typesize) ? "#define XACC(a, b, c) __builtin_fma((a), (b), (c))\n" : "#define XACC(a, b, c) __builtin_fmaf((a), (b), (c))\n";
  else t += "#pragma clang fp contract(off)\n#define XACC(a, b, c) ((c) + (a) * (b))\n"; // the statement as written: multiply, then add
  return t;
}

std::string spgemm_signature(const char* name)
{ // the reference's (A, B, C) plus the batch: element strides of the dense operand and of C between items
  return std::string("extern \"C\" __global__ __launch_bounds__(256) void ") + name
       + "(const T* __restrict__ A, const T* __restrict__ B, T* __restrict__ C, long long stride_dense, long long stride_c, long long batch)\n{\n";
}

// descriptor -> which of the three kernels (generator_spgemm.c:55-145), with the reference's leading-dimension checks
unsigned classify(const libxsmm_gemm_descriptor& d, bool csr, SpKind& kind)
{
  if (0 == d.lda && 0 < d.ldb && 0 < d.ldc) { // A is sparse
    if (csr) { if (d.ldb < d.n) return ERR_LDB; if (d.ldc < d.n) return ERR_LDC; kind = SP_CSR_ASPARSE; }
    else { if (d.ldb < d.k) return ERR_LDB; if (d.ldc < d.m) return ERR_LDC; kind = SP_CSC_ASPARSE; }
    return 0;
  }
  if (0 < d.lda && 0 == d.ldb && 0 < d.ldc) { // B is sparse
    if (csr) return ERR_SPGEMM_GEN; // "B sparse for CSR data structure is not yet available" (:124-127)
    if (d.lda < d.m) return ERR_LDA;
    if (d.ldc < d.m) return ERR_LDC;
    kind = SP_CSC_BSPARSE;
    return 0;
  }
  return ERR_SPGEMM_GEN;
}

SpShape shape_of(const libxsmm_gemm_descriptor& d)
{
  SpShape s;
  s.typesize = (LIBXSMM_GEMM_PRECISION_F64 == LIBXSMM_GETENUM_INP(d.datatype)) ? 8 : 4;
  s.m = (int)d.m; s.n = (int)d.n; s.k = (int)d.k; s.lda = (int)d.lda; s.ldb = (int)d.ldb; s.ldc = (int)d.ldc;
  s.beta0 = (0 != (d.flags & LIBXSMM_GEMM_FLAG_BETA_0)) ? 1 : 0;
  s.v = 0;
  return s;
}

bool supported_precision(const libxsmm_gemm_descriptor& d)
{
  const int ip = LIBXSMM_GETENUM_INP(d.datatype), op = LIBXSMM_GETENUM_OUT(d.datatype);
  return ip == op && (LIBXSMM_GEMM_PRECISION_F64 == ip || LIBXSMM_GEMM_PRECISION_F32 == ip);
}

int fma_default()
{
  const char* const e = getenv("LIBXSMM_AMD_SPGEMM_FMA");
  return (nullptr == e || 0 == *e) ? 1 : (0 != atoi(e) ? 1 : 0);
}

void emit_sparse(libxsmm_generated_code* io, const libxsmm_gemm_descriptor* d, bool csr, const unsigned* row_idx, const unsigned* column_idx)
{
  if (nullptr == io) return;
  if (nullptr == d || nullptr == row_idx || nullptr == column_idx) { fail(io, ERR_SPGEMM_GEN); return; }
  if (!supported_precision(*d)) { fail(io, ERR_UNSUP_DATATYPE); return; }
  SpKind kind = SP_CSR_ASPARSE;
  const unsigned e = classify(*d, csr, kind);
  if (0 != e) { fail(io, e); return; }
  // CSR: row_idx is the row-pointer array, column_idx the column of each entry; CSC: column_idx is the column-pointer
  // array, row_idx the row of each entry (argument naming of the reference, src/generator_spgemm.c:55-60)
  append(io, spgemm_body(kind, shape_of(*d), csr ? row_idx : column_idx, csr ? column_idx : row_idx));
}

// plain dense kernel text for descriptors outside the specialised template's domain (any leading dimensions):
// one thread per C element, k ascending
std::string dense_plain_source(const libxsmm_gemm_descriptor& d, const char* name)
{
  const int ts = (LIBXSMM_GEMM_PRECISION_F64 == LIBXSMM_GETENUM_INP(d.datatype)) ? 8 : 4;
  std::string t = "// generated by libxsmm-amd (dense SMM kernel, plain form)\n";
  t += std::string("typedef ") + tname(ts) + " T;\n";
  t += std::string("extern \"C\" __global__ __launch_bounds__(256) void ") + name
     + "(const T* __restrict__ A, const T* __restrict__ B, T* __restrict__ C, long long stride_a, long long stride_b, long long stride_c, long long batch)\n{\n";
  t += fmt("  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;\n  const long long item = gid / %u;\n", d.m * d.n);
  t += fmt("  const int e = (int)(gid - item * %u), m = e %% %u, n = e / %u;\n  if (item >= batch) return;\n", d.m * d.n, d.m, d.m);
  t += "  const T* const a = A + item * stride_a; const T* const b = B + item * stride_b; T* const c = C + item * stride_c;\n";
  t += fmt("  T acc = %s;\n", (0 != (d.flags & LIBXSMM_GEMM_FLAG_BETA_0)) ? "(T)0" : fmt("c[n * %u + m]", d.ldc).c_str());
  t += fmt("  for (int k = 0; k < %u; ++k) acc = %s(a[k * %u + m], %s, acc);\n", d.k, 8 == ts ? "__builtin_fma" : "__builtin_fmaf", d.lda,
           (0 != (d.flags & LIBXSMM_GEMM_FLAG_TRANS_B)) ? fmt("b[k * %u + n]", d.ldb).c_str() : fmt("b[n * %u + k]", d.ldb).c_str());
  t += fmt("  c[n * %u + m] = acc;\n}\n", d.ldc);
  return t;
}

std::string dense_source(const libxsmm_gemm_descriptor& d, const char* name, unsigned* err)
{
  *err = 0;
  if (!supported_precision(d)) { *err = ERR_UNSUP_DATATYPE; return ""; }
  if (0 != (d.flags & LIBXSMM_GEMM_FLAG_TRANS_A)) { *err = ERR_INVALID_GEMM_CONFIG; return ""; }
  if (d.lda < d.m) { *err = ERR_LDA; return ""; }
  if (d.ldb < ((0 != (d.flags & LIBXSMM_GEMM_FLAG_TRANS_B)) ? d.n : d.k)) { *err = ERR_LDB; return ""; }
  if (d.ldc < d.m) { *err = ERR_LDC; return ""; }
  const int ts = (LIBXSMM_GEMM_PRECISION_F64 == LIBXSMM_GETENUM_INP(d.datatype)) ? 8 : 4;
  const bool tight = (d.lda == d.m && d.ldc == d.m && d.ldb == ((0 != (d.flags & LIBXSMM_GEMM_FLAG_TRANS_B)) ? d.n : d.k));
  if (tight && d.m <= 32 && d.n <= 32 && d.k <= 64) { // the wave-per-item kernel the engine itself launches for large batches
    std::string src = gen_smm_source(ts, (int)d.m, (int)d.n, (int)d.k, d.flags, 0);
    const std::string from = "xsmm_smm_op";
    for (size_t pos = src.find(from); std::string::npos != pos; pos = src.find(from, pos + strlen(name))) src.replace(pos, from.size(), name);
    return src;
  }
  return dense_plain_source(d, name);
}

void write_or_die(const char* file_out, const std::string& text, const char* who)
{ // the reference's file front doors terminate the process on failure (src/generator_spgemm.c:430-446)
  FILE* const f = (nullptr != file_out ? fopen(file_out, "a") : nullptr);
  if (nullptr == f) { fprintf(stderr, "LIBXSMM ERROR: %s could not write to into destination source file\n", who); exit(-1); }
  // a file meant for hipcc needs the runtime header (hiprtc, which compiles the in-memory texts, has it built in):
  // it goes right after the "generated by" line of the translation unit
  const size_t eol = text.find('\n');
  if (std::string::npos != eol && 0 == text.compare(0, 2, "//")) {
    fwrite(text.data(), 1, eol + 1, f);
    fputs("#include <hip/hip_runtime.h>\n", f);
    fputs(text.c_str() + eol + 1, f);
  }
  else fputs(text.c_str(), f);
  fclose(f);
}

} // namespace

LIBXSMM_API const char* libxsmm_strerror(unsigned int i_error_code)
{
  static thread_local char buffer[160];
  const char* msg = nullptr;
  switch (i_error_code) {
    case ERR_GENERAL: msg = "a general error occurred"; break;
    case ERR_ALLOC: msg = "memory allocation failed"; break;
    case ERR_BUFFER_TOO_SMALL: msg = "code buffer too small"; break;
    case ERR_APPEND_STR: msg = "text cannot be appended to a binary code buffer"; break;
    case ERR_ARCH_PREC: case ERR_ARCH: case ERR_UNSUP_ARCH: msg = "unknown or unsupported architecture/precision"; break;
    case ERR_LDA: msg = "lda is too small"; break;
    case ERR_LDB: msg = "ldb is too small"; break;
    case ERR_LDC: msg = "ldc is too small"; break;
    case ERR_SPGEMM_GEN: msg = "could not determine which sparse code generation variant is requested"; break;
    case ERR_CSC_INPUT: msg = "could not open the CSC input file"; break;
    case ERR_CSC_READ_LEN: msg = "line of the CSC file exceeds the line buffer"; break;
    case ERR_CSC_READ_DESC: msg = "header of the CSC file could not be read"; break;
    case ERR_CSC_READ_ELEMS: msg = "element of the CSC file could not be read"; break;
    case ERR_CSC_LEN: msg = "number of elements read differs from the CSC file's header"; break;
    case ERR_CSC_ALLOC_DATA: msg = "could not allocate the CSC data structure"; break;
    case ERR_CSR_ALLOC_DATA: msg = "could not allocate the CSR data structure"; break;
    case ERR_CSR_INPUT: msg = "could not open the CSR input file"; break;
    case ERR_CSR_READ_LEN: msg = "line of the CSR file exceeds the line buffer"; break;
    case ERR_CSR_READ_DESC: msg = "header of the CSR file could not be read"; break;
    case ERR_CSR_READ_ELEMS: msg = "element of the CSR file could not be read"; break;
    case ERR_CSR_LEN: msg = "number of elements read differs from the CSR file's header"; break;
    case ERR_UNSUP_DATATYPE: msg = "unsupported datatype"; break;
    case ERR_INVALID_GEMM_CONFIG: msg = "invalid GEMM configuration"; break;
    case ERR_UNIQUE_VAL: msg = "more unique values than the register kernel can hold"; break;
    default: msg = "unknown error or warning occurred"; break;
  }
  snprintf(buffer, sizeof(buffer), " LIBXSMM ERROR: %s (error #%u)!", msg, i_error_code);
  return buffer;
}

LIBXSMM_API void libxsmm_generator_spgemm_csr_kernel(libxsmm_generated_code* io_generated_code, const libxsmm_gemm_descriptor* i_xgemm_desc,
  const char* i_arch, const unsigned int* i_row_idx, const unsigned int* i_column_idx, const double* i_values)
{
  (void)i_arch; (void)i_values; // one target (gfx950); the values are a run-time argument of the emitted kernel
  emit_sparse(io_generated_code, i_xgemm_desc, true, i_row_idx, i_column_idx);
}

LIBXSMM_API void libxsmm_generator_spgemm_csc_kernel(libxsmm_generated_code* io_generated_code, const libxsmm_gemm_descriptor* i_xgemm_desc,
  const char* i_arch, const unsigned int* i_row_idx, const unsigned int* i_column_idx, const double* i_values)
{
  (void)i_arch; (void)i_values;
  emit_sparse(io_generated_code, i_xgemm_desc, false, i_row_idx, i_column_idx);
}

LIBXSMM_API void libxsmm_generator_spgemm_csr_reg_kernel(libxsmm_generated_code* io_generated_code, const libxsmm_gemm_descriptor* i_xgemm_desc,
  const char* i_arch, const unsigned int* i_row_idx, const unsigned int* i_column_idx, const double* i_values)
{ // values baked in as immediates, rows without non-zeros untouched (src/generator_spgemm_csr_asparse_reg.c:80-313);
  // emits a complete translation unit (kernel "xsmm_csr_op" over column panels), the text libxsmm_create_?csr_reg compiles
  (void)i_arch;
  libxsmm_generated_code* const io = io_generated_code;
  if (nullptr == io) return;
  if (nullptr == i_xgemm_desc || nullptr == i_row_idx || nullptr == i_column_idx || nullptr == i_values) { fail(io, ERR_SPGEMM_GEN); return; }
  const libxsmm_gemm_descriptor& d = *i_xgemm_desc;
  if (!supported_precision(d)) { fail(io, ERR_UNSUP_DATATYPE); return; }
  if (!(0 == d.lda && 0 < d.ldb && 0 < d.ldc)) { fail(io, ERR_SPGEMM_GEN); return; }
  if (d.ldb < d.n) { fail(io, ERR_LDB); return; }
  if (d.ldc < d.n) { fail(io, ERR_LDC); return; }
  const SpShape s = shape_of(d);
  append(io, gen_csr_panels_source(s.typesize, s.m, s.k, i_row_idx, i_column_idx, i_values, s.beta0, 1/*skip empty rows*/, 1, "xsmm_csr_op"));
}

LIBXSMM_API void libxsmm_generator_gemm_kernel(libxsmm_generated_code* io_generated_code, const libxsmm_gemm_descriptor* i_xgemm_desc, const char* i_arch)
{ // dense kernel as HIP text (a complete translation unit; kernel name "xsmm_smm_op")
  (void)i_arch;
  if (nullptr == io_generated_code) return;
  if (nullptr == i_xgemm_desc) { fail(io_generated_code, ERR_INVALID_GEMM_CONFIG); return; }
  unsigned err = 0;
  const std::string src = dense_source(*i_xgemm_desc, "xsmm_smm_op", &err);
  if (0 != err) { fail(io_generated_code, err); return; }
  append(io_generated_code, src);
}

LIBXSMM_API void libxsmm_generator_gemm_inlineasm(const char* i_file_out, const char* i_routine_name, const libxsmm_gemm_descriptor* i_xgemm_desc, const char* i_arch)
{ // appends the kernel text to a source file (the reference emits a C function with inline assembly, src/generator_gemm.c:294-330)
  (void)i_arch;
  unsigned err = (nullptr != i_xgemm_desc && nullptr != i_routine_name) ? 0u : (unsigned)ERR_INVALID_GEMM_CONFIG;
  std::string src;
  if (0 == err) src = dense_source(*i_xgemm_desc, i_routine_name, &err);
  if (0 != err) { fprintf(stderr, "%s\n", libxsmm_strerror(err)); exit(-1); }
  write_or_die(i_file_out, src, "libxsmm_generator_gemm_inlineasm");
}

LIBXSMM_API void libxsmm_generator_gemm_directasm(const char* i_file_out, const char* i_routine_name, const libxsmm_gemm_descriptor* i_xgemm_desc, const char* i_arch)
{ // one textual form on this target: same output as libxsmm_generator_gemm_inlineasm
  libxsmm_generator_gemm_inlineasm(i_file_out, i_routine_name, i_xgemm_desc, i_arch);
}

LIBXSMM_API void libxsmm_generator_spgemm(const char* i_file_out, const char* i_routine_name, const libxsmm_gemm_descriptor* i_xgemm_desc,
  const char* i_arch, const char* i_file_in, const int i_is_csr)
{ // i_is_csr: 0 = CSC file, 1 = CSR file (unrolled text kernel), 3 = CSR file (values baked in, "register" kernel)
  libxsmm_generated_code code; memset(&code, 0, sizeof(code));
  std::vector<unsigned> ptr, idx; std::vector<double> values;
  unsigned rows = 0, cols = 0, nnz = 0, err = 0;
  if (nullptr == i_xgemm_desc || nullptr == i_routine_name) err = ERR_SPGEMM_GEN;
  // i_is_csr: 0 CSC text, 1 CSR text, 2 CSR SOA, 3 CSR with baked-in values, > 9 CSC SOA (src/generator_spgemm.c:271,400-409)
  else if (0 > i_is_csr || (3 < i_is_csr && 10 > i_is_csr)) err = ERR_SPGEMM_GEN;
  const bool file_is_csr = (1 <= i_is_csr && i_is_csr <= 3);
  if (0 == err) err = read_coordinate_file(i_file_in, file_is_csr, ptr, idx, values, rows, cols, nnz);
  if (0 == err) {
    if (3 == i_is_csr) libxsmm_generator_spgemm_csr_reg_kernel(&code, i_xgemm_desc, i_arch, ptr.data(), idx.data(), values.data());
    else {
      const int ts = (LIBXSMM_GEMM_PRECISION_F64 == LIBXSMM_GETENUM_INP(i_xgemm_desc->datatype)) ? 8 : 4;
      append(&code, spgemm_prologue(ts, (2 == i_is_csr || 9 < i_is_csr) ? 1 : fma_default()) + spgemm_signature(i_routine_name));
      if (1 == i_is_csr) libxsmm_generator_spgemm_csr_kernel(&code, i_xgemm_desc, i_arch, ptr.data(), idx.data(), values.data());
      else if (2 == i_is_csr) libxsmm_generator_spgemm_csr_soa_kernel(&code, i_xgemm_desc, i_arch, ptr.data(), idx.data(), values.data());
      else if (0 == i_is_csr) libxsmm_generator_spgemm_csc_kernel(&code, i_xgemm_desc, i_arch, idx.data(), ptr.data(), values.data());
      else libxsmm_generator_spgemm_csc_soa_kernel(&code, i_xgemm_desc, i_arch, idx.data(), ptr.data(), values.data());
      append(&code, "}\n\n");
    }
    err = code.last_error;
  }
  if (0 != err) { // as in the reference: report and terminate (:424-428)
    fprintf(stderr, "%s\n", libxsmm_strerror(err));
    free(code.generated_code);
    exit(-1);
  }
  std::string text(static_cast<const char*>(code.generated_code), code.code_size);
  free(code.generated_code);
  if (3 == i_is_csr) { // the panel kernel's fixed name -> the requested routine name
    const std::string from = "xsmm_csr_op";
    for (size_t pos = text.find(from); std::string::npos != pos; pos = text.find(from, pos + strlen(i_routine_name))) text.replace(pos, from.size(), i_routine_name);
  }
  write_or_die(i_file_out, text, "libxsmm_generator_spgemm");
}

// The MatrixMarket readers behind libxsmm_generator_spgemm as a call of their own (the reference's libxsmm_sparse_csr_reader /
// libxsmm_sparse_csc_reader, src/generator_spgemm_csr_reader.c:46-170, src/generator_spgemm_csc_reader.c:46-170, are
// internal to its generator library; its samples carry copies, samples/edge/common_edge_proxy.h:50-280).
LIBXSMM_API int libxsmm_amd_sparse_reader(const char* path, int is_csr, unsigned int** o_ptr, unsigned int** o_idx, double** o_values,
  unsigned int* o_row_count, unsigned int* o_column_count, unsigned int* o_element_count)
{
  if (nullptr == o_ptr || nullptr == o_idx || nullptr == o_values || nullptr == o_row_count || nullptr == o_column_count || nullptr == o_element_count) return (int)ERR_SPGEMM_GEN;
  *o_ptr = nullptr; *o_idx = nullptr; *o_values = nullptr; *o_row_count = *o_column_count = *o_element_count = 0;
  std::vector<unsigned> ptr, idx; std::vector<double> values;
  unsigned rows = 0, cols = 0, nnz = 0;
  const unsigned err = read_coordinate_file(path, 0 != is_csr, ptr, idx, values, rows, cols, nnz);
  if (0 != err) return (int)err;
  unsigned* const p = static_cast<unsigned*>(malloc(sizeof(unsigned) * ptr.size()));
  unsigned* const i = static_cast<unsigned*>(malloc(sizeof(unsigned) * (idx.empty() ? 1 : idx.size())));
  double* const v = static_cast<double*>(malloc(sizeof(double) * (values.empty() ? 1 : values.size())));
  if (nullptr == p || nullptr == i || nullptr == v) { free(p); free(i); free(v); return (int)(0 != is_csr ? ERR_CSR_ALLOC_DATA : ERR_CSC_ALLOC_DATA); }
  memcpy(p, ptr.data(), sizeof(unsigned) * ptr.size());
  if (!idx.empty()) memcpy(i, idx.data(), sizeof(unsigned) * idx.size());
  if (!values.empty()) memcpy(v, values.data(), sizeof(double) * values.size());
  *o_ptr = p; *o_idx = i; *o_values = v; *o_row_count = rows; *o_column_count = cols; *o_element_count = nnz;
  return 0;
}

// ---- executable form -----------------------------------------------------------------------------------------------
struct libxsmm_amd_spgemm {
  JitKernel* kernel;
  SpKind kind;
  SpShape shape;
  unsigned nnz;
  int lanes;
};

LIBXSMM_API libxsmm_amd_spgemm* libxsmm_amd_spgemm_create(const libxsmm_gemm_descriptor* descriptor, int is_csr,
  const unsigned int* row_idx, const unsigned int* column_idx, int fma)
{
  if (nullptr == descriptor || nullptr == row_idx || nullptr == column_idx || !supported_precision(*descriptor)) return nullptr;
  SpKind kind = SP_CSR_ASPARSE;
  if (0 != classify(*descriptor, 0 != is_csr, kind)) return nullptr;
  if (!device_ready()) { fail_no_device("libxsmm_amd_spgemm_create"); return nullptr; }
  const SpShape s = shape_of(*descriptor);
  const unsigned* const ptr = (0 != is_csr ? row_idx : column_idx);
  const unsigned* const idx = (0 != is_csr ? column_idx : row_idx);
  const int nmajor = (SP_CSR_ASPARSE == kind ? s.m : (SP_CSC_BSPARSE == kind ? s.n : s.k));
  const std::string src = spgemm_prologue(s.typesize, fma < 0 ? fma_default() : fma) + spgemm_signature("xsmm_spgemm_op") + spgemm_body(kind, s, ptr, idx) + "}\n";
  std::string log;
  JitKernel* const k = jit_compile(src, "xsmm_spgemm_op", &log);
  if (nullptr == k) {
    if (0 != libxsmm_verbosity) fprintf(stderr, "LIBXSMM-AMD ERROR: sparse text kernel did not compile (%s)\n", log.c_str());
    return nullptr;
  }
  libxsmm_amd_spgemm* const h = new libxsmm_amd_spgemm;
  h->kernel = k; h->kind = kind; h->shape = s; h->nnz = ptr[nmajor]; h->lanes = sp_lanes(kind, s);
  return h;
}

LIBXSMM_API void libxsmm_amd_spgemm_destroy(const libxsmm_amd_spgemm* handle)
{
  if (nullptr == handle) return;
  jit_release(handle->kernel);
  delete const_cast<libxsmm_amd_spgemm*>(handle);
}

LIBXSMM_API int libxsmm_amd_spgemm_execute_batch(const libxsmm_amd_spgemm* handle, const void* sparse_values, const void* dense, void* c,
  long long stride_dense, long long stride_c, long long batch)
{
  if (nullptr == handle || nullptr == sparse_values || nullptr == dense || nullptr == c || batch < 0) return EXIT_FAILURE;
  if (0 == batch) return EXIT_SUCCESS;
  if (!device_ready()) { fail_no_device("libxsmm_amd_spgemm_execute_batch"); return EXIT_FAILURE; }
  const SpShape& s = handle->shape;
  const size_t ts = (size_t)s.typesize;
  // elements one item touches in the dense operand and in C
  size_t ext_dense, ext_c;
  switch (handle->kind) {
    case SP_CSR_ASPARSE: ext_dense = (size_t)(s.k - 1) * s.ldb + s.n; ext_c = (size_t)(s.m - 1) * s.ldc + (0 != s.beta0 ? s.ldc : s.n); break;
    case SP_CSC_BSPARSE: ext_dense = (size_t)(s.k - 1) * s.lda + s.m; ext_c = (size_t)(s.n - 1) * s.ldc + s.m; break;
    case SP_SOA_ASPARSE: case SP_SOA_RM_BC: ext_dense = ((size_t)(s.k - 1) * s.ldb + s.n) * s.v; ext_c = ((size_t)(s.m - 1) * s.ldc + s.n) * s.v; break;
    case SP_SOA_BSPARSE: case SP_SOA_RM_AC: ext_dense = ((size_t)(s.m - 1) * s.lda + s.k) * s.v; ext_c = ((size_t)(s.m - 1) * s.ldc + s.n) * s.v; break;
    default: ext_dense = (size_t)(s.n - 1) * s.ldb + s.k; ext_c = (size_t)(s.n - 1) * s.ldc + s.m; break;
  }
  const size_t bytes_dense = ((size_t)(batch - 1) * (size_t)stride_dense + ext_dense) * ts;
  const size_t bytes_c = ((size_t)(batch - 1) * (size_t)stride_c + ext_c) * ts;
  const void* dv = sparse_values; const void* dd = dense; void* dc = c;
  const bool c_host = !is_device_ptr(c);
  if (!is_device_ptr(sparse_values)) { void* t = scratch(0, (size_t)handle->nnz * ts); if (nullptr == t || 0 != h2d(t, sparse_values, (size_t)handle->nnz * ts)) return EXIT_FAILURE; dv = t; }
  if (!is_device_ptr(dense)) { void* t = scratch(3, bytes_dense); if (nullptr == t || 0 != h2d(t, dense, bytes_dense)) return EXIT_FAILURE; dd = t; }
  if (c_host) { void* t = scratch(5, bytes_c); if (nullptr == t || 0 != h2d(t, c, bytes_c)) return EXIT_FAILURE; dc = t; }
  // argument order of the emitted kernel: (A, B, C, ...) with the sparse operand's values in A (A sparse) or B (B sparse)
  const bool dense_is_a = (SP_CSC_BSPARSE == handle->kind || SP_SOA_BSPARSE == handle->kind || SP_SOA_RM_AC == handle->kind);
  const void* pa = (dense_is_a ? dd : dv);
  const void* pb = (dense_is_a ? dv : dd);
  const long long threads = batch * handle->lanes;
  const long long blocks = (threads + 255) / 256;
  if (blocks > 0x7fffffffLL) return EXIT_FAILURE;
  void* args[] = { (void*)&pa, (void*)&pb, (void*)&dc, (void*)&stride_dense, (void*)&stride_c, (void*)&batch };
  const int e = jit_launch_args(handle->kernel, (unsigned)blocks, 256u, args, device().stream);
  static const char* const names[] = { "spgemm_csr_asparse_text", "spgemm_csc_bsparse_text", "spgemm_csc_asparse_text",
                                       "soa_asparse_text", "soa_bsparse_text", "soa_rm_ac_text", "soa_rm_bc_text" };
  note_launch(names[handle->kind]);
  if (0 != e) return EXIT_FAILURE;
  if (c_host) return 0 == d2h(c, dc, bytes_c) ? EXIT_SUCCESS : EXIT_FAILURE;
  if (dv != sparse_values || dd != dense) return 0 == stream_sync() ? EXIT_SUCCESS : EXIT_FAILURE; // staged inputs must have landed
  settle(sparse_values, dense, c);
  return EXIT_SUCCESS;
}

LIBXSMM_API int libxsmm_amd_spgemm_source(const libxsmm_gemm_descriptor* descriptor, int is_csr, const unsigned int* row_idx,
  const unsigned int* column_idx, int fma, char* buffer, size_t buffer_size, int compile)
{ // the text libxsmm_amd_spgemm_create compiles; conventions of libxsmm_amd_csr_kernel_source
  if (nullptr == descriptor || nullptr == row_idx || nullptr == column_idx || !supported_precision(*descriptor)) return -1;
  SpKind kind = SP_CSR_ASPARSE;
  if (0 != classify(*descriptor, 0 != is_csr, kind)) return -1;
  const SpShape s = shape_of(*descriptor);
  const std::string src = spgemm_prologue(s.typesize, fma < 0 ? fma_default() : fma) + spgemm_signature("xsmm_spgemm_op")
                        + spgemm_body(kind, s, 0 != is_csr ? row_idx : column_idx, 0 != is_csr ? column_idx : row_idx) + "}\n";
  if (nullptr != buffer && 0 < buffer_size) {
    const size_t n = (src.size() < buffer_size - 1 ? src.size() : buffer_size - 1);
    memcpy(buffer, src.data(), n); buffer[n] = 0;
  }
  if (0 != compile) {
    std::string log;
    const int rc = jit_check_source(src, &log);
    if (0 != rc && 0 != libxsmm_verbosity) fprintf(stderr, "LIBXSMM-AMD: hiprtc: %s\n", log.c_str());
    return rc;
  }
  return (int)src.size();
}

// ---- SOA family ------------------------------------------------------------------------------------------------------
namespace {

// descriptor -> SOA kernel kind with the reference's checks (src/generator_spgemm.c:179-238); form: 0 CSR, 1 CSC, 2 rm_ac, 3 rm_bc
unsigned classify_soa(const libxsmm_gemm_descriptor& d, int form, SpKind& kind)
{
  if (2 == form || 3 == form) {
    if (d.lda < d.k) return ERR_LDA;
    if (d.ldb < d.n) return ERR_LDB;
    if (d.ldc < d.n) return ERR_LDC;
    kind = (2 == form ? SP_SOA_RM_AC : SP_SOA_RM_BC);
    return 0;
  }
  if (0 == form && 0 == d.lda && 0 < d.ldb && 0 < d.ldc) { // A sparse (CSR only)
    if (d.ldb < d.n) return ERR_LDB;
    if (d.ldc < d.n) return ERR_LDC;
    kind = SP_SOA_ASPARSE;
    return 0;
  }
  if (0 < d.lda && 0 == d.ldb && 0 < d.ldc) { // B sparse (CSR or CSC)
    if (d.lda < d.k) return ERR_LDA;
    if (d.ldc < d.n) return ERR_LDC;
    kind = SP_SOA_BSPARSE;
    return 0;
  }
  return ERR_SPGEMM_GEN;
}

int soa_width(int typesize) { return 8 == typesize ? 8 : 16; }

// indexes must address the operator: the generated statements are unconditional
bool soa_pattern_ok(SpKind kind, const SpShape& s, const unsigned* ptr, const unsigned* idx, bool csr)
{
  if (SP_SOA_RM_AC == kind || SP_SOA_RM_BC == kind) return true;
  if (nullptr == ptr || nullptr == idx) return false;
  const int nmajor = (SP_SOA_ASPARSE == kind) ? s.m : (csr ? s.k : s.n);
  const unsigned limit = (unsigned)((SP_SOA_ASPARSE == kind) ? s.k : (csr ? s.n : s.k));
  for (int i = 0; i < nmajor; ++i) if (ptr[i] > ptr[i + 1]) return false;
  if (SP_SOA_ASPARSE == kind) for (unsigned p = ptr[0]; p < ptr[nmajor]; ++p) if (idx[p] >= limit) return false;
  return true;
}

std::string soa_source(SpKind kind, const SpShape& s, const unsigned* ptr, const unsigned* idx, bool csr, const char* name)
{
  return spgemm_prologue(s.typesize, 1/*fma*/) + spgemm_signature(name) + soa_body(kind, s, ptr, idx, csr) + "}\n";
}

libxsmm_amd_spgemm* soa_create(const libxsmm_gemm_descriptor* descriptor, int form, const unsigned* ptr, const unsigned* idx)
{
  if (nullptr == descriptor || !supported_precision(*descriptor)) return nullptr;
  SpKind kind = SP_SOA_ASPARSE;
  if (0 != classify_soa(*descriptor, form, kind)) return nullptr;
  SpShape s = shape_of(*descriptor);
  s.v = soa_width(s.typesize);
  const bool csr = (0 == form);
  if (0 == s.m || 0 == s.n || 0 == s.k || !soa_pattern_ok(kind, s, ptr, idx, csr)) return nullptr;
  if (!device_ready()) { fail_no_device("libxsmm_create_*_soa"); return nullptr; }
  std::string log;
  JitKernel* const k = jit_compile(soa_source(kind, s, ptr, idx, csr, "xsmm_spgemm_op"), "xsmm_spgemm_op", &log);
  if (nullptr == k) {
    if (0 != libxsmm_verbosity) fprintf(stderr, "LIBXSMM-AMD ERROR: SOA kernel did not compile (%s)\n", log.c_str());
    return nullptr;
  }
  libxsmm_amd_spgemm* const h = new libxsmm_amd_spgemm;
  h->kernel = k; h->kind = kind; h->shape = s; h->lanes = sp_lanes(kind, s);
  switch (kind) { // elements of the shared operand (staged when it lives on the host)
    case SP_SOA_ASPARSE: h->nnz = ptr[s.m]; break;
    case SP_SOA_BSPARSE: h->nnz = ptr[csr ? s.k : s.n]; break;
    case SP_SOA_RM_AC: h->nnz = (unsigned)((s.k - 1) * s.ldb + s.n); break;
    default: h->nnz = (unsigned)((s.m - 1) * s.lda + s.k); break;
  }
  return h;
}

libxsmm_xmmfunction soa_kernel(const libxsmm_gemm_descriptor* descriptor, int form, const unsigned* ptr, const unsigned* idx)
{
  libxsmm_xmmfunction result; result.xmm = nullptr;
  libxsmm_init();
  libxsmm_amd_spgemm* const h = soa_create(descriptor, form, ptr, idx);
  if (nullptr == h) return result;
  Kernel* const k = new Kernel();
  k->desc = *descriptor; k->kclass = KC_TEXT; k->text = h;
  void* const fn = adopt_kernel(k);
  if (nullptr == fn) { libxsmm_amd_spgemm_destroy(h); delete k; return result; }
  result.xmm = reinterpret_cast<decltype(result.xmm)>(fn);
  return result;
}

void emit_soa(libxsmm_generated_code* io, const libxsmm_gemm_descriptor* d, int form, const unsigned* ptr, const unsigned* idx)
{
  if (nullptr == io) return;
  if (nullptr == d || nullptr == ptr || nullptr == idx) { fail(io, ERR_SPGEMM_GEN); return; }
  if (!supported_precision(*d)) { fail(io, ERR_UNSUP_DATATYPE); return; }
  SpKind kind = SP_SOA_ASPARSE;
  const unsigned e = classify_soa(*d, form, kind);
  if (0 != e) { fail(io, e); return; }
  SpShape s = shape_of(*d); s.v = soa_width(s.typesize);
  if (!soa_pattern_ok(kind, s, ptr, idx, 0 == form)) { fail(io, ERR_SPGEMM_GEN); return; }
  append(io, soa_body(kind, s, ptr, idx, 0 == form));
}

} // namespace

namespace xsmm {
int text_kernel_execute(void* text, const void* a, const void* b, void* c, long long stride_dense, long long stride_c, long long batch)
{ // kernel(a, b, c): which of a/b is the shared operand follows from the kind
  const libxsmm_amd_spgemm* const h = static_cast<const libxsmm_amd_spgemm*>(text);
  if (nullptr == h) return EXIT_FAILURE;
  const bool dense_is_a = (SP_CSC_BSPARSE == h->kind || SP_SOA_BSPARSE == h->kind || SP_SOA_RM_AC == h->kind);
  return libxsmm_amd_spgemm_execute_batch(h, dense_is_a ? b : a, dense_is_a ? a : b, c, stride_dense, stride_c, batch);
}
void text_kernel_destroy(void* text) { libxsmm_amd_spgemm_destroy(static_cast<const libxsmm_amd_spgemm*>(text)); }
}

LIBXSMM_API libxsmm_xmmfunction libxsmm_create_xcsr_soa(const libxsmm_gemm_descriptor* descriptor,
  const unsigned int* row_ptr, const unsigned int* column_idx, const void* values)
{ // src/libxsmm_main.c:2423-2447; `values` only has to be non-NULL (the kernel takes the values at call time)
  libxsmm_xmmfunction none; none.xmm = nullptr;
  if (nullptr == descriptor || nullptr == row_ptr || nullptr == column_idx || nullptr == values) return none;
  return soa_kernel(descriptor, 0, row_ptr, column_idx);
}

LIBXSMM_API libxsmm_xmmfunction libxsmm_create_xcsc_soa(const libxsmm_gemm_descriptor* descriptor,
  const unsigned int* column_ptr, const unsigned int* row_idx, const void* values)
{ // src/libxsmm_main.c:2450-2474
  libxsmm_xmmfunction none; none.xmm = nullptr;
  if (nullptr == descriptor || nullptr == column_ptr || nullptr == row_idx || nullptr == values) return none;
  return soa_kernel(descriptor, 1, column_ptr, row_idx);
}

LIBXSMM_API libxsmm_xmmfunction libxsmm_create_rm_ac_soa(const libxsmm_gemm_descriptor* descriptor)
{ // src/libxsmm_main.c:2477-2497
  return soa_kernel(descriptor, 2, nullptr, nullptr);
}

LIBXSMM_API libxsmm_xmmfunction libxsmm_create_rm_bc_soa(const libxsmm_gemm_descriptor* descriptor)
{ // src/libxsmm_main.c:2500-2520
  return soa_kernel(descriptor, 3, nullptr, nullptr);
}

LIBXSMM_API void libxsmm_generator_spgemm_csr_soa_kernel(libxsmm_generated_code* io_generated_code, const libxsmm_gemm_descriptor* i_xgemm_desc,
  const char* i_arch, const unsigned int* i_row_idx, const unsigned int* i_column_idx, const void* i_values)
{
  (void)i_arch; (void)i_values;
  emit_soa(io_generated_code, i_xgemm_desc, 0, i_row_idx, i_column_idx);
}

LIBXSMM_API void libxsmm_generator_spgemm_csc_soa_kernel(libxsmm_generated_code* io_generated_code, const libxsmm_gemm_descriptor* i_xgemm_desc,
  const char* i_arch, const unsigned int* i_row_idx, const unsigned int* i_column_idx, const void* i_values)
{ // argument naming of the reference: i_row_idx = row of each entry, i_column_idx = column pointers
  (void)i_arch; (void)i_values;
  emit_soa(io_generated_code, i_xgemm_desc, 1, i_column_idx, i_row_idx);
}

LIBXSMM_API int libxsmm_amd_soa_width(libxsmm_gemm_precision precision)
{
  return LIBXSMM_GEMM_PRECISION_F64 == precision ? 8 : (LIBXSMM_GEMM_PRECISION_F32 == precision ? 16 : 0);
}

LIBXSMM_API int libxsmm_amd_kernel_execute_batch(const void* kernel, const void* a, const void* b, void* c,
  long long stride_dense, long long stride_c, long long batch)
{
  Kernel* const k = kernel_from_pointer(kernel);
  if (nullptr == k || KC_TEXT != k->kclass) return EXIT_FAILURE;
  return text_kernel_execute(k->text, a, b, c, stride_dense, stride_c, batch);
}
