// spmv_coo.hip -- COO -> CSR on the device: the reference's convert_in_csr
// (src/csr_matrix.c:63-126: histogram, scan, scatter in file order, per-row quicksort on the host)
// as upload of the triplets + one stable radix sort by (row, column) + two small kernels.  The sort
// is rocPRIM's (a ROCm library routine; nothing on the SpMV path goes through it).
#include "spmv_internal.hpp"

#include <rocprim/rocprim.hpp>

namespace {

// key = row << 32 | column; out-of-range indices are reported through *bad (first offender wins)
__global__ __launch_bounds__(kBlock) void coo_make_keys(long long nz, int M, int N, const int *__restrict__ I,
                                                        const int *__restrict__ J, unsigned long long *__restrict__ key,
                                                        unsigned long long *__restrict__ bad) {
    const long long e = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (e >= nz) return;
    const int r = I[e], c = J[e];
    if ((unsigned)r >= (unsigned)M || (unsigned)c >= (unsigned)N) atomicMin(bad, (unsigned long long)e);
    key[e] = ((unsigned long long)(unsigned)r << 32) | (unsigned)c;
}

// sorted keys -> col[e] and row_ptr (rows without entries included)
__global__ __launch_bounds__(kBlock) void coo_split_keys(long long nz, int M, const unsigned long long *__restrict__ key,
                                                         int *__restrict__ col, int *__restrict__ row_ptr) {
    const long long e = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (e > nz) return;
    if (e == nz) {  // rows behind the last entry's row (all rows when nz == 0)
        const int last = nz ? (int)(key[nz - 1] >> 32) : -1;
        for (int r = last + 1; r <= M; ++r) row_ptr[r] = (int)nz;
        return;
    }
    const unsigned long long k = key[e];
    col[e] = (int)(unsigned)k;
    const int r = (int)(k >> 32);
    const int prev = e ? (int)(key[e - 1] >> 32) : -1;
    for (int q = prev + 1; q <= r; ++q) row_ptr[q] = (int)e;  // usually zero or one iteration
}

}  // namespace

// COO triplets (0-based, any order, file order = tie order) -> a CSR handle, built on the device.
// Equal to convert_in_csr + spmv_hip_csr_upload except for the order of entries that repeat the same
// (row, column): the stable sort keeps them in file order, the reference's quicksort does not.
static int spmv_hip_csr_from_coo_body(int M, int N, long long nz, const int *I, const int *J, const double *val,
                                     spmv_csr_dev **out) {
    if (need_device()) return -1;
    if (!out) return fail("csr_from_coo: out is NULL");
    *out = nullptr;
    if (M < 0 || N < 0 || nz < 0 || nz > 0x7fffffffLL - kPad || (nz > 0 && (!I || !J || !val)))
        return fail("csr_from_coo: bad arguments");
    int *d_I = nullptr, *d_J = nullptr, *d_col = nullptr, *d_rp = nullptr;
    double *d_vin = nullptr, *d_val = nullptr;
    unsigned long long *d_kin = nullptr, *d_kout = nullptr, *d_bad = nullptr;
    void *d_tmp = nullptr;
    std::vector<int> rp((size_t)M + 1, 0);
    int rc = -1;
    do {
        const size_t n = (size_t)nz, n1 = std::max<size_t>(n, 1);
        hipError_t e = hipMalloc((void **)&d_I, n1 * sizeof(int));
        if (e == hipSuccess) e = hipMalloc((void **)&d_J, n1 * sizeof(int));
        if (e == hipSuccess) e = hipMalloc((void **)&d_vin, n1 * sizeof(double));
        if (e == hipSuccess) e = hipMalloc((void **)&d_kin, n1 * sizeof(unsigned long long));
        if (e == hipSuccess) e = hipMalloc((void **)&d_kout, n1 * sizeof(unsigned long long));
        if (e == hipSuccess) e = hipMalloc((void **)&d_val, (n + kPad) * sizeof(double));
        if (e == hipSuccess) e = hipMalloc((void **)&d_col, (n + kPad) * sizeof(int));
        if (e == hipSuccess) e = hipMalloc((void **)&d_rp, ((size_t)M + 1) * sizeof(int));
        if (e == hipSuccess) e = hipMalloc((void **)&d_bad, sizeof(unsigned long long));
        if (e == hipSuccess) e = hipMemsetAsync(d_bad, 0xFF, sizeof(unsigned long long), g_stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_val + n, 0, kPad * sizeof(double), g_stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_col + n, 0, kPad * sizeof(int), g_stream);
        if (e == hipSuccess && n) e = hipMemcpyAsync(d_I, I, n * sizeof(int), hipMemcpyHostToDevice, g_stream);
        if (e == hipSuccess && n) e = hipMemcpyAsync(d_J, J, n * sizeof(int), hipMemcpyHostToDevice, g_stream);
        if (e == hipSuccess && n) e = hipMemcpyAsync(d_vin, val, n * sizeof(double), hipMemcpyHostToDevice, g_stream);
        if (e != hipSuccess) { fail("csr_from_coo: allocation / upload failed: %s", hipGetErrorString(e)); break; }
        const int grid = (int)((nz + kBlock) / kBlock);  // covers e == nz as well
        if (n) {
            hipLaunchKernelGGL(coo_make_keys, dim3(grid), dim3(kBlock), 0, g_stream, nz, M, N, d_I, d_J, d_kin, d_bad);
            // only the bits that can differ: 32 of the column + those of the row
            unsigned row_bits = 1;
            while (row_bits < 32 && (1ull << row_bits) < (unsigned long long)std::max(M, 1)) ++row_bits;
            size_t tmp_bytes = 0;
            e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_kin, d_kout, d_vin, d_val, n, 0u, 32u + row_bits, g_stream);
            if (e == hipSuccess) e = hipMalloc(&d_tmp, std::max<size_t>(tmp_bytes, 16));
            if (e == hipSuccess)
                e = rocprim::radix_sort_pairs(d_tmp, tmp_bytes, d_kin, d_kout, d_vin, d_val, n, 0u, 32u + row_bits, g_stream);
            if (e != hipSuccess) { fail("csr_from_coo: sort failed: %s", hipGetErrorString(e)); break; }
        }
        hipLaunchKernelGGL(coo_split_keys, dim3(grid), dim3(kBlock), 0, g_stream, nz, M, d_kout, d_col, d_rp);
        unsigned long long bad = 0;
        e = hipMemcpyAsync(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost, g_stream);
        if (e == hipSuccess) e = hipMemcpyAsync(rp.data(), d_rp, rp.size() * sizeof(int), hipMemcpyDeviceToHost, g_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
        if (e != hipSuccess) { fail("csr_from_coo: build failed: %s", hipGetErrorString(e)); break; }
        if (bad != ~0ull) {
            fail("csr_from_coo: entry %llu (%d, %d) is outside the %d x %d matrix", bad, I[bad], J[bad], M, N);
            break;
        }
        rc = csr_adopt_f64(M, N, rp.data(), d_col, d_val, out);
        if (rc == 0) d_col = nullptr, d_val = nullptr;  // the handle owns them now
    } while (0);
    (void)hipFree(d_I);
    (void)hipFree(d_J);
    (void)hipFree(d_vin);
    (void)hipFree(d_kin);
    (void)hipFree(d_kout);
    (void)hipFree(d_rp);
    (void)hipFree(d_bad);
    (void)hipFree(d_tmp);
    (void)hipFree(d_col);
    (void)hipFree(d_val);
    return rc;
}

extern "C" int spmv_hip_csr_from_coo(int M, int N, long long nz, const int *I, const int *J, const double *val,
                                     spmv_csr_dev **out) {
    return guarded("csr_from_coo", [&] { return spmv_hip_csr_from_coo_body(M, N, nz, I, J, val, out); });
}

// The CSR arrays of a handle back to the host (tests; hosts that let the device build the matrix):
// row_ptr[M_local + 1] (rebased to 0), col[nz], val[nz] of the handle's dtype; any pointer may be NULL.
extern "C" int spmv_hip_csr_download(const spmv_csr_dev *m, int *row_ptr, int *col, void *val) {
    if (need_device()) return -1;
    if (!m) return fail("csr_download: NULL handle");
    HIP_TRY(hipStreamSynchronize(g_stream));
    if (row_ptr) HIP_TRY(hipMemcpy(row_ptr, m->row_ptr, ((size_t)m->M_local + 1) * sizeof(int), hipMemcpyDeviceToHost));
    if (col && m->nz) HIP_TRY(hipMemcpy(col, m->col, (size_t)m->nz * sizeof(int), hipMemcpyDeviceToHost));
    if (val && m->nz) HIP_TRY(hipMemcpy(val, m->val, (size_t)m->nz * m->value_bytes, hipMemcpyDeviceToHost));
    return 0;
}
