// C-ABI plumbing shared by all kernel files: error string, device probe, stream-ordered memset/copy helpers.
#include <hip/hip_runtime.h>
#include <string.h>

static thread_local char g_err[512] = "";

extern "C" void padt_set_error(const char* msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}

extern "C" const char* padt_last_error(void) { return g_err; }

extern "C" int padt_abi_version(void) { return 4; }
// common.h PADT_STREAM_SCALE of the two operand-type instantiations
extern "C" float padt_stream_scale(int f16) { return f16 ? 0.0625f : 1.0f; }

// Returns 0 and fills name (<= 255 chars) / CU count when a gfx950 device is present; -1 otherwise.
extern "C" int padt_device_info(int device, char* name, int name_len, int* n_cu, long* hbm_bytes) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { padt_set_error("padt_device_info: no HIP device"); return -1; }
    if (name && name_len > 0) { strncpy(name, prop.gcnArchName, name_len - 1); name[name_len - 1] = 0; }
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (long)prop.totalGlobalMem;
    return 0;
}

extern "C" int padt_memset(void* stream, void* dst, int value, long bytes) {
    if (bytes <= 0) return 0;
    hipError_t e = hipMemsetAsync(dst, value, (size_t)bytes, (hipStream_t)stream);
    if (e != hipSuccess) { padt_set_error(hipGetErrorString(e)); return -2; }
    return 0;
}

// Device-side timing on an arbitrary stream (bench.py's roofline leg: torch.cuda.Event only sees torch's stream).
extern "C" int padt_event_create(void** ev) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) { padt_set_error("hipEventCreate failed"); return -2; }
    *ev = (void*)e;
    return 0;
}
extern "C" int padt_event_record(void* ev, void* stream) {
    return hipEventRecord((hipEvent_t)ev, (hipStream_t)stream) == hipSuccess ? 0 : -2;
}
extern "C" int padt_event_elapsed_ms(void* start, void* stop, float* ms) {
    if (hipEventSynchronize((hipEvent_t)stop) != hipSuccess) return -2;
    return hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop) == hipSuccess ? 0 : -2;
}
extern "C" int padt_event_destroy(void* ev) { return hipEventDestroy((hipEvent_t)ev) == hipSuccess ? 0 : -2; }
