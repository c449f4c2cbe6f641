// One-way latency of a flag hand-off between two workgroups, by ping-pong (no clock comparison between XCDs): workgroup 0 and
// workgroup `peer` bounce a counter 1000 times; time / 2000 = store -> seen by the other side's poll.  Variants: plain word in
// the same L2 (peer on the same XCD), agent-scope word (sc1) across XCDs, with a release fence in front / an acquire behind.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/handoff scripts/ubench/handoff.hip && /tmp/handoff
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>   // 0: relaxed agent store / load;  1: + release fence before the store, acquire fence after the wait;  2: + s_sleep 2 between polls
__global__ void k(int* flag, long long* t, int peer, int n) {
    if (blockIdx.x != 0 && blockIdx.x != (unsigned)peer) return;
    if (threadIdx.x != 0) return;
    const bool a = blockIdx.x == 0;
    int* mine = flag + (a ? 0 : 32);
    int* theirs = flag + (a ? 32 : 0);
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 1; i <= n; ++i) {
        if (a) {
            if (MODE >= 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_store(mine, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < i) { if (MODE == 2) __builtin_amdgcn_s_sleep(2); }
            if (MODE >= 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        } else {
            while (__hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < i) { if (MODE == 2) __builtin_amdgcn_s_sleep(2); }
            if (MODE >= 1) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); }
            __hip_atomic_store(mine, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (a) t[0] = __builtin_amdgcn_s_memrealtime() - t0;
}
template <int MODE> void run(const char* nm, int peer) {
    int* f; long long* t; (void)hipMalloc(&f, 512); (void)hipMalloc(&t, 8);
    long long h = 0; const int n = 1000;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipMemset(f, 0, 512);
        hipLaunchKernelGGL(k<MODE>, dim3(16), dim3(64), 0, 0, f, t, peer, n); (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    printf("%-72s peer workgroup %2d: %6.2f us one way\n", nm, peer, (double)h / 100.0 / (2.0 * n));
}
int main() {
    run<0>("agent-scope relaxed store / poll", 1);
    run<0>("agent-scope relaxed store / poll", 8);
    run<1>("+ release fence before the store, acquire after the wait", 1);
    run<1>("+ release fence before the store, acquire after the wait", 8);
    run<2>("+ s_sleep 2 between polls", 1);
    run<2>("+ s_sleep 2 between polls", 8);
    return 0;
}
