// Round 3, verdict item 7 (raw AQL launch path for small batches): what a raw AQL dispatch on an own HSA queue saves against
// hipModuleLaunchKernel, measured on the smallest kernel with the library's completion scheme (a ticket in pinned host memory the
// host spins on): time from the launch call / the doorbell to the ticket's arrival, back to back.
// usage (GPU box): ./aql_probe ticket_kernel.hsaco [iterations]
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <vector>

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define HSACHK(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { const char *m_ = nullptr; hsa_status_string(s_, &m_); std::fprintf(stderr, "%s: %s\n", #x, m_ ? m_ : "?"); return 3; } } while (0)

static hsa_agent_t g_gpu; static bool g_have_gpu = false;
static hsa_status_t find_gpu(hsa_agent_t a, void *) {
    hsa_device_type_t t;
    if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) == HSA_STATUS_SUCCESS && t == HSA_DEVICE_TYPE_GPU && !g_have_gpu) { g_gpu = a; g_have_gpu = true; }
    return HSA_STATUS_SUCCESS;
}
static hsa_amd_memory_pool_t g_kernarg_pool; static bool g_have_pool = false;
static hsa_status_t find_kernarg_pool(hsa_amd_memory_pool_t p, void *) {
    hsa_amd_segment_t seg; uint32_t flags = 0;
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
    if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    if ((flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_KERNARG_INIT) && !g_have_pool) { g_kernarg_pool = p; g_have_pool = true; }
    return HSA_STATUS_SUCCESS;
}
static hsa_agent_t g_cpu; static bool g_have_cpu = false;
static hsa_status_t find_cpu(hsa_agent_t a, void *) {
    hsa_device_type_t t;
    if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) == HSA_STATUS_SUCCESS && t == HSA_DEVICE_TYPE_CPU && !g_have_cpu) { g_cpu = a; g_have_cpu = true; }
    return HSA_STATUS_SUCCESS;
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static bool spin(volatile unsigned long long *flag, unsigned long long want) {
    const double t0 = now_us();
    while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != want) {
        __builtin_ia32_pause();
        if (now_us() - t0 > 2e6) return false;   // (two seconds: something is wrong, give up instead of hanging the box)
    }
    return true;
}
static void stats(const char *what, std::vector<double> &v) {
    std::sort(v.begin(), v.end());
    std::printf("%-44s p10 %6.2f  p50 %6.2f  p90 %6.2f us  (n = %zu)\n", what, v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10], v.size());
}

int main(int argc, char **argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: aql_probe ticket_kernel.hsaco [iterations]\n"); return 1; }
    const int iters = argc > 2 ? std::atoi(argv[2]) : 2000;
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<char> image((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (image.empty()) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }

    // ---- HIP side: module launch + ticket
    HIPCHK(hipSetDevice(0));
    unsigned long long *flag = nullptr, *flag_dev = nullptr;
    HIPCHK(hipHostMalloc((void **)&flag, 64, hipHostMallocDefault));
    HIPCHK(hipHostGetDevicePointer((void **)&flag_dev, flag, 0));
    *flag = 0;
    hipModule_t mod; hipFunction_t fn;
    HIPCHK(hipModuleLoadData(&mod, image.data()));
    HIPCHK(hipModuleGetFunction(&fn, mod, "ticket_kernel"));
    hipStream_t st;
    HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned long long seq = 0;
    std::vector<double> t_hip, t_hip_call;
    for (int i = 0; i < iters + 200; ++i) {
        ++seq;
        struct { unsigned long long *p; unsigned long long s; } args = {flag_dev, seq};
        size_t sz = sizeof(args);
        void *cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
        const double t0 = now_us();
        HIPCHK(hipModuleLaunchKernel(fn, 1, 1, 1, 64, 1, 1, 0, st, nullptr, cfg));
        const double t1 = now_us();
        if (!spin(flag, seq)) { std::fprintf(stderr, "HIP launch %d: no ticket\n", i); return 4; }
        const double t2 = now_us();
        if (i >= 200) { t_hip.push_back(t2 - t0); t_hip_call.push_back(t1 - t0); }
    }
    HIPCHK(hipStreamSynchronize(st));
    stats("hipModuleLaunchKernel -> ticket", t_hip);
    stats("  of it inside the launch call", t_hip_call);

    // ---- raw AQL on an own queue
    HSACHK(hsa_init());
    HSACHK(hsa_iterate_agents(find_gpu, nullptr));
    HSACHK(hsa_iterate_agents(find_cpu, nullptr));
    if (!g_have_gpu || !g_have_cpu) { std::fprintf(stderr, "no agents\n"); return 3; }
    HSACHK(hsa_amd_agent_iterate_memory_pools(g_cpu, find_kernarg_pool, nullptr));
    if (!g_have_pool) { std::fprintf(stderr, "no kernarg pool\n"); return 3; }
    hsa_queue_t *q = nullptr;
    HSACHK(hsa_queue_create(g_gpu, 256, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q));
    hsa_code_object_reader_t reader;
    HSACHK(hsa_code_object_reader_create_from_memory(image.data(), image.size(), &reader));
    hsa_executable_t exe;
    HSACHK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe));
    HSACHK(hsa_executable_load_agent_code_object(exe, g_gpu, reader, nullptr, nullptr));
    HSACHK(hsa_executable_freeze(exe, nullptr));
    hsa_executable_symbol_t sym;
    HSACHK(hsa_executable_get_symbol_by_name(exe, "ticket_kernel.kd", &g_gpu, &sym));
    uint64_t kobj = 0; uint32_t karg = 0, group = 0, priv = 0;
    HSACHK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &kobj));
    HSACHK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &karg));
    HSACHK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &group));
    HSACHK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &priv));
    std::printf("kernel object %#llx, kernarg %u B, group %u B, private %u B\n", (unsigned long long)kobj, karg, group, priv);
    if (priv != 0) { std::fprintf(stderr, "the probe kernel must not need scratch\n"); return 3; }
    const size_t slot = std::max<size_t>(256, (karg + 63) & ~63u);
    char *kargs = nullptr;   // a ring of kernarg slots, one per queue entry
    HSACHK(hsa_amd_memory_pool_allocate(g_kernarg_pool, slot * 256, 0, (void **)&kargs));
    HSACHK(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, kargs));
    std::memset(kargs, 0, slot * 256);
    std::vector<double> t_aql, t_aql_call;
    const uint32_t mask = q->size - 1;
    for (int i = 0; i < iters + 200; ++i) {
        ++seq;
        const double t0 = now_us();
        const uint64_t idx = hsa_queue_add_write_index_relaxed(q, 1);
        while (idx - hsa_queue_load_read_index_scacquire(q) >= q->size) __builtin_ia32_pause();
        char *ka = kargs + slot * (idx & 255);
        struct { unsigned long long *p; unsigned long long s; } args = {flag_dev, seq};
        std::memcpy(ka, &args, sizeof(args));
        hsa_kernel_dispatch_packet_t *pk = reinterpret_cast<hsa_kernel_dispatch_packet_t *>(q->base_address) + (idx & mask);
        pk->workgroup_size_x = 64; pk->workgroup_size_y = 1; pk->workgroup_size_z = 1; pk->reserved0 = 0;
        pk->grid_size_x = 64; pk->grid_size_y = 1; pk->grid_size_z = 1;
        pk->private_segment_size = 0; pk->group_segment_size = group;
        pk->kernel_object = kobj; pk->kernarg_address = ka; pk->reserved2 = 0;
        pk->completion_signal.handle = 0;
        const uint16_t header = (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) |
                                           (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                                           (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
        const uint16_t setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
        __atomic_store_n(reinterpret_cast<uint32_t *>(pk), (uint32_t)header | ((uint32_t)setup << 16), __ATOMIC_RELEASE);
        hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)idx);
        const double t1 = now_us();
        if (!spin(flag, seq)) { std::fprintf(stderr, "AQL dispatch %d: no ticket\n", i); return 4; }
        const double t2 = now_us();
        if (i >= 200) { t_aql.push_back(t2 - t0); t_aql_call.push_back(t1 - t0); }
    }
    stats("raw AQL packet + doorbell -> ticket", t_aql);
    stats("  of it writing the packet", t_aql_call);
    hsa_queue_destroy(q);
    hsa_executable_destroy(exe);
    hsa_code_object_reader_destroy(reader);
    hsa_amd_memory_pool_free(kargs);
    return 0;
}
