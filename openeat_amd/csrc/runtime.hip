// Error plumbing and ABI version of libopeneat_hip.so.
#include "oe_common.h"
#include "../../include/openeat_hip.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

extern "C" void oe_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* oe_last_error(void) { return g_err; }

extern "C" int oe_abi_version(void) { return 1; }

// ---- capture hygiene ------------------------------------------------------------------------------------------------
// A stream capture that forks work onto other streams must lead every fork back into the origin stream before
// hipStreamEndCapture, or the capture is invalid ("unjoined work"; on this ROCm a five-stream capture that ended that way
// took the process down with SIGSEGV inside capture_end instead of returning the error).  This walks the graph under
// construction: the nodes a side stream's next launch would depend on must all be ancestors (or members) of the set the
// origin's next launch would depend on.  Host-only, no launches; called by TrainEngine.capture right before it ends the
// capture.
#include <unordered_set>
#include <vector>
extern "C" int oe_capture_unjoined_streams(void* origin, void* const* sides, int n_sides, int* unjoined_index) {
    OE_REQUIRE(origin && (n_sides == 0 || sides), "oe_capture_unjoined_streams: null pointer");
    hipStreamCaptureStatus st;
    unsigned long long id0 = 0;
    hipGraph_t graph = nullptr;
    const hipGraphNode_t* deps = nullptr;
    size_t ndeps = 0;
    hipError_t e = hipStreamGetCaptureInfo_v2((hipStream_t)origin, &st, &id0, &graph, &deps, &ndeps);
    OE_REQUIRE(e == hipSuccess, "oe_capture_unjoined_streams: hipStreamGetCaptureInfo_v2: %s", hipGetErrorString(e));
    OE_REQUIRE(st == hipStreamCaptureStatusActive, "oe_capture_unjoined_streams: the origin stream is not capturing");
    std::unordered_set<hipGraphNode_t> seen;
    std::vector<hipGraphNode_t> todo(deps, deps + ndeps), buf;
    for (hipGraphNode_t d : todo) seen.insert(d);
    while (!todo.empty()) {
        hipGraphNode_t nd = todo.back();
        todo.pop_back();
        size_t k = 0;
        e = hipGraphNodeGetDependencies(nd, nullptr, &k);
        OE_REQUIRE(e == hipSuccess, "oe_capture_unjoined_streams: hipGraphNodeGetDependencies: %s", hipGetErrorString(e));
        if (k == 0) continue;
        buf.resize(k);
        e = hipGraphNodeGetDependencies(nd, buf.data(), &k);
        OE_REQUIRE(e == hipSuccess, "oe_capture_unjoined_streams: hipGraphNodeGetDependencies: %s", hipGetErrorString(e));
        for (size_t i = 0; i < k; ++i)
            if (seen.insert(buf[i]).second) todo.push_back(buf[i]);
    }
    int bad = 0;
    for (int s = 0; s < n_sides; ++s) {
        if (!sides[s] || sides[s] == origin) continue;
        hipStreamCaptureStatus ss;
        unsigned long long id = 0;
        const hipGraphNode_t* sd = nullptr;
        size_t nsd = 0;
        e = hipStreamGetCaptureInfo_v2((hipStream_t)sides[s], &ss, &id, nullptr, &sd, &nsd);
        OE_REQUIRE(e == hipSuccess, "oe_capture_unjoined_streams: hipStreamGetCaptureInfo_v2(side %d): %s", s, hipGetErrorString(e));
        if (ss != hipStreamCaptureStatusActive || id != id0) continue;       // never forked into this capture
        bool joined = true;
        for (size_t i = 0; i < nsd; ++i)
            if (!seen.count(sd[i])) joined = false;
        if (!joined) {
            if (bad == 0 && unjoined_index) *unjoined_index = s;
            ++bad;
        }
    }
    return bad;
}


// ---- phase stamps (diagnostic: tools/phase_stamps.py) ------------------------------------------------------------------
// One thread writes the constant-rate wall clock (100 MHz) into slot `slot` of `buf` when the stream reaches this point:
// captured into the step's graph it gives the un-profiled timeline of the replay at the chosen points (a kernel trace
// changes what overlaps).  Never launched by the product path unless OE_PHASE_STAMPS is set.
__global__ void stamp_kernel(long long* buf, int slot) { buf[slot] = wall_clock64(); }
extern "C" int oe_stamp(long long* buf, int slot, void* stream) {
    OE_REQUIRE(buf && slot >= 0, "oe_stamp: bad arguments");
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, buf, slot);
    OE_LAUNCH_CHECK("stamp");
    return 0;
}
