// microprobe (round 4, experiment dropped - DESIGN 8): the segmented suffix sum of the multi-level flush with DPP row shifts against the ds_bpermute form
// on the same inputs.  As written it shows the trap: the run flags `n16 + d < 16 && row_up<d>(key) == key` evaluate the exchange under the short-circuit,
// and a lane the short-circuit has switched off reads as 0 to its neighbours (DPP and ds_bpermute alike): 12 lanes get a wrong flag.
// hipcc --offload-arch=gfx950 -O3 ab/micro/dpp_probe2.hip -o ab/micro/dpp_probe2
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
template <int S> __device__ __forceinline__ uint32_t row_up(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x100 + S, 0xf, 0xf, true); }   // lane l gets lane l + S of its 16-lane row
template <int S> __device__ __forceinline__ float row_up(float v) { return __builtin_bit_cast(float, row_up<S>(__builtin_bit_cast(uint32_t, v))); }
__device__ __forceinline__ float seg_dpp(float v, const bool (&same)[4]) {
    float pv = row_up<1>(v); v += same[0] ? pv : 0.f;
    pv = row_up<2>(v); v += same[1] ? pv : 0.f;
    pv = row_up<4>(v); v += same[2] ? pv : 0.f;
    pv = row_up<8>(v); v += same[3] ? pv : 0.f;
    return v;
}
__device__ __forceinline__ float seg_shfl(float v, const bool (&same)[4], int ln) {
    for (int k = 0; k < 4; ++k) { const float pv = __shfl(v, ln + (1 << k)); v += same[k] ? pv : 0.f; }
    return v;
}
__global__ void k(const uint32_t* keys, const float* vals, float* out) {
    const int ln = threadIdx.x, n16 = ln & 15;
    const uint32_t key = keys[ln];
    const bool sd[4] = {n16 + 1 < 16 && row_up<1>(key) == key, n16 + 2 < 16 && row_up<2>(key) == key, n16 + 4 < 16 && row_up<4>(key) == key, n16 + 8 < 16 && row_up<8>(key) == key};
    bool ss[4];
    for (int k2 = 0; k2 < 4; ++k2) { const int d = 1 << k2; const uint32_t pk = (uint32_t)__shfl((int)key, ln + d); ss[k2] = n16 + d < 16 && pk == key; }
    float a[3], b[3];
    for (int i = 0; i < 3; ++i) { a[i] = seg_dpp(vals[64 * i + ln], sd); b[i] = seg_shfl(vals[64 * i + ln], ss, ln); }
    for (int i = 0; i < 3; ++i) { out[64 * i + ln] = a[i]; out[192 + 64 * i + ln] = b[i]; }
    out[384 + ln] = (float)((sd[0] != ss[0]) + (sd[1] != ss[1]) + (sd[2] != ss[2]) + (sd[3] != ss[3]));
}
int main() {
    uint32_t hk[64]; float hv[192], ho[448];
    for (int i = 0; i < 64; ++i) hk[i] = 7 + (i & 15) / 4 + 100 * (i >> 4);          // runs of 4 lanes
    for (int i = 0; i < 192; ++i) hv[i] = 1.0f + 0.01f * i;
    uint32_t* dk; float *dv, *dout; hipMalloc(&dk, 256); hipMalloc(&dv, 768); hipMalloc(&dout, 448 * 4);
    hipMemcpy(dk, hk, 256, hipMemcpyHostToDevice); hipMemcpy(dv, hv, 768, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dk, dv, dout);
    hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 192; ++i) bad += ho[i] != ho[192 + i];
    int badflag = 0; for (int i = 0; i < 64; ++i) badflag += ho[384 + i] != 0.f;
    printf("mismatching sums: %d of 192; lanes with mismatching run flags: %d\n", bad, badflag);
    for (int i = 0; i < 64; ++i) if (ho[384 + i] != 0.f || ho[i] != ho[192 + i]) printf("lane %2d key %u dpp %.3f shfl %.3f flagdiff %.0f\n", i, hk[i], ho[i], ho[192 + i], ho[384 + i]);
    return 0;
}
