// Runtime plumbing of libfcnhip.so: device selection, memory, streams, events, hipGraph capture.
// Stands in for what pycaffe hides behind caffe.set_device/set_mode_gpu and SyncedMemory
// (reference call sites: scripts/fcn_object_detector.py:68-69,82,87-90).
#include <stdarg.h>

#include "common.h"

namespace fcn {

char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int set_err(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace fcn

using namespace fcn;

extern "C" {

int fcn_abi_version(void) { return FCN_ABI_VERSION; }

const char* fcn_last_error_string(void) { return err_buf(); }

int fcn_device_count(int* count) {
    FCN_REQUIRE(count, FCN_E_ARG, "fcn_device_count: null");
    FCN_HIP(hipGetDeviceCount(count));
    return 0;
}

int fcn_init(int device) {
    FCN_HIP(hipSetDevice(device));
    int rc = 0;
    zero_page_for_current_device(&rc);   // first call per device allocates; later calls are a table lookup
    return rc;
}

int fcn_device_name(char* h_buf, int len) {
    FCN_REQUIRE(h_buf && len > 0, FCN_E_ARG, "fcn_device_name: bad buffer");
    int dev = 0;
    FCN_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    FCN_HIP(hipGetDeviceProperties(&prop, dev));
    snprintf(h_buf, len, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return 0;
}

int fcn_device_sync(void) {
    FCN_HIP(hipDeviceSynchronize());
    return 0;
}

int fcn_malloc(void** p, size_t bytes) {
    FCN_REQUIRE(p, FCN_E_ARG, "fcn_malloc: null");
    if (bytes == 0) bytes = 16;
    FCN_HIP(hipMalloc(p, bytes));
    return 0;
}

int fcn_free(void* p) {
    if (p) FCN_HIP(hipFree(p));
    return 0;
}

int fcn_host_malloc(void** h_p, size_t bytes) {
    FCN_REQUIRE(h_p, FCN_E_ARG, "fcn_host_malloc: null");
    if (bytes == 0) bytes = 16;
    FCN_HIP(hipHostMalloc(h_p, bytes, hipHostMallocDefault));
    return 0;
}

int fcn_host_free(void* h_p) {
    if (h_p) FCN_HIP(hipHostFree(h_p));
    return 0;
}

int fcn_memset_async(void* p, int value, size_t bytes, fcn_stream_t s) {
    FCN_REQUIRE(p || bytes == 0, FCN_E_ARG, "fcn_memset_async: null");
    if (bytes) FCN_HIP(hipMemsetAsync(p, value, bytes, as_stream(s)));
    return 0;
}

int fcn_memcpy_h2d_async(void* dst, const void* h_src, size_t bytes, fcn_stream_t s) {
    FCN_REQUIRE((dst && h_src) || bytes == 0, FCN_E_ARG, "fcn_memcpy_h2d_async: null");
    if (bytes) FCN_HIP(hipMemcpyAsync(dst, h_src, bytes, hipMemcpyHostToDevice, as_stream(s)));
    return 0;
}

int fcn_memcpy_d2h_async(void* h_dst, const void* src, size_t bytes, fcn_stream_t s) {
    FCN_REQUIRE((h_dst && src) || bytes == 0, FCN_E_ARG, "fcn_memcpy_d2h_async: null");
    if (bytes) FCN_HIP(hipMemcpyAsync(h_dst, src, bytes, hipMemcpyDeviceToHost, as_stream(s)));
    return 0;
}

int fcn_memcpy_d2d_async(void* dst, const void* src, size_t bytes, fcn_stream_t s) {
    FCN_REQUIRE((dst && src) || bytes == 0, FCN_E_ARG, "fcn_memcpy_d2d_async: null");
    if (bytes) FCN_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, as_stream(s)));
    return 0;
}

int fcn_stream_create(fcn_stream_t* s) {
    FCN_REQUIRE(s, FCN_E_ARG, "fcn_stream_create: null");
    hipStream_t st;
    FCN_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    *s = st;
    return 0;
}

int fcn_stream_destroy(fcn_stream_t s) {
    if (s) FCN_HIP(hipStreamDestroy(as_stream(s)));
    return 0;
}

int fcn_stream_sync(fcn_stream_t s) {
    FCN_HIP(hipStreamSynchronize(as_stream(s)));
    return 0;
}

int fcn_event_create(fcn_event_t* e) {
    FCN_REQUIRE(e, FCN_E_ARG, "fcn_event_create: null");
    hipEvent_t ev;
    FCN_HIP(hipEventCreate(&ev));
    *e = ev;
    return 0;
}

int fcn_event_destroy(fcn_event_t e) {
    if (e) FCN_HIP(hipEventDestroy((hipEvent_t)e));
    return 0;
}

int fcn_event_record(fcn_event_t e, fcn_stream_t s) {
    FCN_REQUIRE(e, FCN_E_ARG, "fcn_event_record: null");
    FCN_HIP(hipEventRecord((hipEvent_t)e, as_stream(s)));
    return 0;
}

int fcn_event_sync(fcn_event_t e) {
    FCN_REQUIRE(e, FCN_E_ARG, "fcn_event_sync: null");
    FCN_HIP(hipEventSynchronize((hipEvent_t)e));
    return 0;
}

int fcn_event_elapsed_ms(fcn_event_t start, fcn_event_t stop, float* h_ms) {
    FCN_REQUIRE(start && stop && h_ms, FCN_E_ARG, "fcn_event_elapsed_ms: null");
    FCN_HIP(hipEventElapsedTime(h_ms, (hipEvent_t)start, (hipEvent_t)stop));
    return 0;
}

int fcn_stream_wait_event(fcn_stream_t s, fcn_event_t e) {
    FCN_REQUIRE(e, FCN_E_ARG, "fcn_stream_wait_event: null event");
    FCN_HIP(hipStreamWaitEvent(as_stream(s), (hipEvent_t)e, 0));
    return 0;
}

int fcn_graph_begin(fcn_stream_t s) {
    FCN_REQUIRE(s, FCN_E_ARG, "fcn_graph_begin: capture needs an explicit stream");
    FCN_HIP(hipStreamBeginCapture(as_stream(s), hipStreamCaptureModeThreadLocal));
    return 0;
}

int fcn_graph_end(fcn_stream_t s, fcn_graph_t* g) {
    FCN_REQUIRE(s && g, FCN_E_ARG, "fcn_graph_end: null");
    hipGraph_t graph = nullptr;
    FCN_HIP(hipStreamEndCapture(as_stream(s), &graph));
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) return set_err(-(int)e, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
    *g = exec;
    return 0;
}

int fcn_graph_launch(fcn_graph_t g, fcn_stream_t s) {
    FCN_REQUIRE(g, FCN_E_ARG, "fcn_graph_launch: null");
    FCN_HIP(hipGraphLaunch((hipGraphExec_t)g, as_stream(s)));
    return 0;
}

int fcn_graph_destroy(fcn_graph_t g) {
    if (g) FCN_HIP(hipGraphExecDestroy((hipGraphExec_t)g));
    return 0;
}

}  // extern "C"
