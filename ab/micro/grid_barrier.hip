// microbenchmark: what does a device-scope barrier of a persistent grid cost on MI355X (8 XCDs, one L2 each)?  The question behind "one launch per training
// step" (VERDICT r03 item 3): a step needs two grid-wide barriers (all gradient flushes and records visible -> reduce + Adam; new parameters visible -> next
// step), each with an agent-scope release / acquire pair (L2 write-back + invalidate across XCDs).  256 workgroups x 512 threads, one per CU like the fused
// kernels; every iteration each workgroup writes BYTES of its own record, passes the barrier, and reads a slice of EVERY workgroup's record (the reduce
// pattern).  The spin is bounded: a logic error ends in wrong numbers, not in a hung GPU.
// (diagnostic only; build: hipcc --offload-arch=gfx950 -O3 ab/micro/grid_barrier.hip -o ab/micro/grid_barrier)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

// SCOPE 0: C11 fences without a scope (= system scope in HIP: the L2 is written back as for the host); 1: agent-scope fences (__builtin_amdgcn_fence)
template <int SCOPE>
__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        if (SCOPE == 0) __atomic_thread_fence(__ATOMIC_RELEASE);
        else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");                            // this workgroup's stores are visible device-wide
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spin = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spin > (1 << 22)) { ok = false; break; }
        }
        if (SCOPE == 0) __atomic_thread_fence(__ATOMIC_ACQUIRE);
        else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return ok;
}

// two-level form: one arrival counter and one release flag per XCD (workgroup b runs on XCD b & 7), the last arriver of an XCD reports to a global counter,
// polls it and releases its XCD: 32 pollers per address instead of 256
__device__ __forceinline__ bool grid_barrier2(unsigned* c, unsigned gen, int nb) {       // c: [0..7] XCD counters (64 B apart), [128] global, [256 + 16 x] release flags
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        const int x = blockIdx.x & 7, per = nb >> 3;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned old = __hip_atomic_fetch_add(&c[16 * x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spin = 0;
        if (old + 1 == gen * per) {                                                        // the last workgroup of this XCD
            __hip_atomic_fetch_add(&c[128], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(&c[128], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen * 8u) {
                __builtin_amdgcn_s_sleep(1);
                if (++spin > (1 << 22)) { ok = false; break; }
            }
            __hip_atomic_store(&c[256 + 16 * x], gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(&c[256 + 16 * x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) {
                __builtin_amdgcn_s_sleep(1);
                if (++spin > (1 << 22)) { ok = false; break; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return ok;
}

template <int FENCES>
__global__ void __launch_bounds__(512) loop(float* rec, int rec_floats, unsigned* counter, float* out, int iters, int* err) {
    float acc = 0.f;
    const int nb = gridDim.x;
    for (int it = 0; it < iters; ++it) {
        float* mine = rec + (size_t)blockIdx.x * rec_floats;
        for (int i = threadIdx.x; i < rec_floats; i += 512) mine[i] = (float)(it + i);
        if (FENCES) {
            if (!(FENCES == 3 ? grid_barrier2(counter, (unsigned)(2 * it + 1), nb) : grid_barrier<FENCES == 3 ? 1 : FENCES - 1>(counter, (unsigned)(nb * (2 * it + 1))))) { if (threadIdx.x == 0) atomicAdd(err, 1); break; }
        }
        // the reduce pattern: this workgroup's slice of every record
        const int per = (rec_floats + nb - 1) / nb;
        for (int j = threadIdx.x; j < per; j += 512) {
            const int o = blockIdx.x * per + j;
            if (o < rec_floats) {
                float part[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};                    // eight loads in flight, like reduce_q16_kernel
                for (int w = 0; w + 8 <= nb; w += 8)
#pragma unroll
                    for (int k = 0; k < 8; ++k) part[k] += __builtin_nontemporal_load(&rec[(size_t)(w + k) * rec_floats + o]);
                acc += ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
            }
        }
        if (FENCES) {
            if (!(FENCES == 3 ? grid_barrier2(counter, (unsigned)(2 * it + 2), nb) : grid_barrier<FENCES == 3 ? 1 : FENCES - 1>(counter, (unsigned)(nb * (2 * it + 2))))) { if (threadIdx.x == 0) atomicAdd(err, 1); break; }
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc;
}

int main() {
    const int nb = 256, iters = 200;
    for (int rec_kb : {0, 8, 80}) {
        const int rf = rec_kb == 0 ? 64 : rec_kb * 256;        // 0: 256 B, barriers all but alone
        float *rec, *out; unsigned* counter; int* err;
        hipMalloc(&rec, (size_t)nb * rf * 4); hipMalloc(&out, nb * 512 * 4); hipMalloc(&counter, 4096); hipMalloc(&err, 4);
        for (int fences = 0; fences < 4; ++fences) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                hipMemset(counter, 0, 4096); hipMemset(err, 0, 4);
                hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
                hipEventRecord(a);
                if (fences == 3) hipLaunchKernelGGL(loop<3>, dim3(nb), dim3(512), 0, 0, rec, rf, counter, out, iters, err);
                else if (fences == 2) hipLaunchKernelGGL(loop<2>, dim3(nb), dim3(512), 0, 0, rec, rf, counter, out, iters, err);
                else if (fences == 1) hipLaunchKernelGGL(loop<1>, dim3(nb), dim3(512), 0, 0, rec, rf, counter, out, iters, err);
                else hipLaunchKernelGGL(loop<0>, dim3(nb), dim3(512), 0, 0, rec, rf, counter, out, iters, err);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            int herr = 0; hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
            printf("record %3d KB per workgroup, %s: %.2f us per iteration (%d iterations, 256 workgroups)%s\n", rec_kb,
                   fences == 3 ? "two TWO-LEVEL barriers (per-XCD counters)" : fences == 2 ? "two barriers, agent-scope fences" : (fences == 1 ? "two barriers, unscoped (system) fences" : "no barrier (racy: timing floor of the traffic)"), best * 1e3f / iters, iters, herr ? "  BARRIER TIMED OUT" : "");
        }
        hipFree(rec); hipFree(out); hipFree(counter); hipFree(err);
    }
    return 0;
}
