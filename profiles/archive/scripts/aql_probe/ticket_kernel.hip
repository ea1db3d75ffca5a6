// The smallest kernel with the completion scheme of the library's epilogue: one lane writes a sequence number into pinned host
// memory (system scope) -- what the host spins on instead of a driver signal.
#include <hip/hip_runtime.h>
extern "C" __global__ void ticket_kernel(unsigned long long *host_flag, unsigned long long seq) {
    if (threadIdx.x == 0) __hip_atomic_store(host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
