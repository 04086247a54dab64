// what does ds_read_b64_tr_b16 deliver?  LDS element i (16-bit) holds the value i; lane L supplies byte address 8*L (case A)
// or the address pattern of a [row][col] block read (case B); print each lane's 4 received elements.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
__global__ void k(int* out, int mode) {
    __shared__ __attribute__((aligned(16))) short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int L = threadIdx.x;
    int elem;                                   // element index this lane points at
    if (mode == 0) elem = 4 * L;                // consecutive 8-byte chunks
    else {                                      // block of 4 rows x 16 cols per 16-lane group, row stride 100 elements
        const int l16 = L & 15, q = l16 >> 2, p = l16 & 3, grp = L >> 4;
        elem = (q + 4 * grp) * 100 + 4 * p;
        if (mode == 2) elem = (4 * (L >> 5) + q) * 72 + 16 * ((L >> 4) & 1) + 4 * p;      // two groups side by side in the same 4 rows
    }
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + elem));
    for (int j = 0; j < 4; ++j) out[L * 4 + j] = v[j];
}
int main() {
    int* d; hipMalloc(&d, 64 * 4 * 4);
    int h[256];
    for (int mode = 2; mode < 3; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int L = 0; L < 64; ++L) printf("L%02d: %5d %5d %5d %5d%s", L, h[4 * L], h[4 * L + 1], h[4 * L + 2], h[4 * L + 3], (L & 3) == 3 ? "\n" : "   ");
    }
    return 0;
}
