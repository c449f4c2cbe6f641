// One-launch Adam step for the training loops of train_utils.py (the reference's torch.optim.Adam(lr=0.1) over a handful
// of tiny parameter tensors: train_utils.py:43,100,166,238,291).  The gradients come flattened in one buffer (the
// caller gathers them with one torch.cat launch: autograd puts them wherever it likes, the parameters stay put).  At the reference's sizes an iteration is launch-bound
//  and torch's capturable Adam is 13 multi-tensor launches per step; this is one.  Same arithmetic as
// torch.optim.Adam with amsgrad = False, weight_decay = 0, maximize = False:
//   m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// The step count t lives on the device (the launch can be replayed from a hipGraph): every workgroup reads it on entry,
// the last one to finish stores t + 1.
#include "common.h"
#include "../../include/volt_hip.h"

namespace volt {

struct AdamSlot {            // one parameter tensor
    float* p;
    float* m;
    float* v;
    long long end;           // one past its last element in the flattened index space
};

__global__ __launch_bounds__(256) void adam_kernel(const AdamSlot* __restrict__ slots, int nslots, long long total,
                                                  const float* __restrict__ grad, float lr, float b1, float b2, float eps, int* __restrict__ state) {
    const int t = state[0] + 1;
    const float bc1 = 1.f - __powf(b1, (float)t), bc2 = 1.f - __powf(b2, (float)t);
    const float step = lr / bc1, rs2 = 1.f / sqrtf(bc2);
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        int s = 0;
        while (s + 1 < nslots && e >= slots[s].end) ++s;
        const long long i = e - (s ? slots[s - 1].end : 0);
        const float g = grad[e];
        const float m = b1 * slots[s].m[i] + (1.f - b1) * g;
        const float v = b2 * slots[s].v[i] + (1.f - b2) * g * g;
        slots[s].m[i] = m;
        slots[s].v[i] = v;
        slots[s].p[i] -= step * m / (sqrtf(v) * rs2 + eps);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // every workgroup has read state[0] before it gets here, so the last arrival may overwrite it
        if (atomicAdd(state + 1, 1) == (int)gridDim.x - 1) {
            state[1] = 0;
            state[0] = t;
        }
    }
}

}  // namespace volt

using namespace volt;

extern "C" int volt_adam_step_f32(const void* slots, int nslots, long long total, const float* grad, float lr, float beta1,
                                  float beta2, float eps, int* state, void* stream) {
    if (!slots) return -1;
    if (nslots < 1) return -2;
    if (total < 1) return -3;
    if (!grad) return -4;
    if (!state) return -9;
    const int grid = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
    hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const AdamSlot*)slots, nslots, total, grad,
                       lr, beta1, beta2, eps, state);
    VOLT_LAUNCH_CHECK();
    return 0;
}
