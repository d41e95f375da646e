// Probe for the peer-to-peer gather: can a second PROCESS open this process's device buffers (hipIpc*, dmabuf mode)
// and are its kernel's stores seen (a) after its kernel ended, through an ordinary hipMalloc buffer, and (b) while a
// bounded spin-wait kernel of the owner is running, through an uncached signal buffer?
//   hipcc --offload-arch=gfx950 -O2 -o ipc_probe ipc_probe.hip && ./ipc_probe
// The child is forked BEFORE any HIP call (no fork/exec after the GPU was initialised).
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <sys/wait.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); _exit(3); } } while (0)

__global__ void fill(float* data, int n, unsigned* flag, unsigned stamp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) data[i] = 1.0f + i;
    if (flag && i == 0) {   // the real path signals from a separate, stream-ordered kernel; this probe does the same below
        __atomic_store_n(flag, stamp, __ATOMIC_RELEASE);
    }
}
__global__ void signal(unsigned* flag, unsigned stamp) {
    __threadfence_system();
    __hip_atomic_store(flag, stamp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void wait_flag(const unsigned* flag, unsigned stamp, unsigned* result, long max_spins) {
    long k = 0;
    unsigned v = 0;
    for (; k < max_spins; ++k) {   // bounded: every wave reaches the end
        v = __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (v >= stamp) break;
        __builtin_amdgcn_s_sleep(8);
    }
    result[0] = v;
    result[1] = k < max_spins ? 1u : 0u;
    result[2] = (unsigned)(k & 0xffffffff);
}

struct Msg { hipIpcMemHandle_t data, flag; };

int main() {
    int to_child[2], to_parent[2];
    if (pipe(to_child) || pipe(to_parent)) return 2;
    const int n = 1 << 16;
    pid_t pid = fork();
    if (pid == 0) {   // child = owner of the buffers (the "learner side")
        CK(hipSetDevice(0));
        float* data; unsigned* flag; unsigned* result;
        CK(hipMalloc(&data, n * sizeof(float)));
        CK(hipMemset(data, 0, n * sizeof(float)));
        CK(hipExtMallocWithFlags((void**)&flag, 256, hipDeviceMallocUncached));
        CK(hipMemset(flag, 0, 256));
        CK(hipMalloc(&result, 64));
        CK(hipDeviceSynchronize());
        Msg m;
        CK(hipIpcGetMemHandle(&m.data, data));
        CK(hipIpcGetMemHandle(&m.flag, flag));
        // spin-wait kernel is already running when the peer writes
        wait_flag<<<1, 1>>>(flag, 7u, result, 5000000L);
        if (write(to_parent[1], &m, sizeof m) != sizeof m) _exit(4);
        CK(hipDeviceSynchronize());
        unsigned r[3];
        CK(hipMemcpy(r, result, sizeof r, hipMemcpyDeviceToHost));
        // a later kernel boundary: rows written by the peer's kernel
        float* host = (float*)malloc(n * sizeof(float));
        CK(hipMemcpy(host, data, n * sizeof(float), hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < n; ++i) bad += host[i] != 1.0f + i;
        printf("owner: flag seen=%u reached=%u spins=%u; rows wrong=%d of %d\n", r[0], r[1], r[2], bad, n);
        char c = (r[1] == 1 && bad == 0) ? 'k' : 'f';
        if (write(to_parent[1], &c, 1) != 1) _exit(4);
        _exit(c == 'k' ? 0 : 1);
    }
    // parent = the peer that writes
    Msg m;
    if (read(to_parent[0], &m, sizeof m) != sizeof m) return 4;
    CK(hipSetDevice(0));
    float* data; unsigned* flag;
    CK(hipIpcOpenMemHandle((void**)&data, m.data, hipIpcMemLazyEnablePeerAccess));
    CK(hipIpcOpenMemHandle((void**)&flag, m.flag, hipIpcMemLazyEnablePeerAccess));
    fill<<<n / 256, 256>>>(data, n, nullptr, 0);
    signal<<<1, 1>>>(flag, 7u);
    CK(hipDeviceSynchronize());
    char c = 0;
    if (read(to_parent[0], &c, 1) != 1) return 4;
    int status = 0;
    waitpid(pid, &status, 0);
    CK(hipIpcCloseMemHandle(data));
    CK(hipIpcCloseMemHandle(flag));
    printf("peer: owner says %c, exit %d\n", c, WEXITSTATUS(status));
    return c == 'k' ? 0 : 1;
}
