// Launch-sequence recorder / executor.
//
// A model forward (SDXL UNet: ~1000 kernel launches) is described ONCE by calling the ordinary
// C-ABI launchers while a program is recording: instead of launching, every launcher stores a
// closure with its arguments.  The recorded program is then replayed from C++ in a single call
// (no per-op Python/ctypes cost) or instantiated as a hipGraph and launched as one graph
// (no per-op host launch cost; ~1.2-1.5 us per kernel boundary on MI355X instead of ~3.5 us of
// host work per launch).  Values that change between replays (sigmas, timesteps, guidance,
// latents, conditioning) live in device buffers that the host refreshes before a replay, so the
// captured graph stays static.
//
// This is the MI355X-native stand-in for the reference's optional stable-fast "compile"
// (CUDA-graph + Triton, /root/reference/latentblending/blending_engine.py:88-96).
#include "lb_common.h"
#include <functional>
#include <string>
#include <vector>

struct LbOp {
    std::string name;
    std::function<int(hipStream_t)> fn;
};

struct LbProgram {
    std::vector<LbOp> ops;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipStream_t capture_stream = nullptr;
};

static thread_local LbProgram* g_recording = nullptr;

bool lb_recording() { return g_recording != nullptr; }

void lb_record(const char* name, std::function<int(hipStream_t)> fn) {
    g_recording->ops.push_back(LbOp{std::string(name), std::move(fn)});
}

extern "C" void* lb_program_create(void) { return new LbProgram(); }

extern "C" void lb_program_destroy(void* prog) {
    LbProgram* p = (LbProgram*)prog;
    if (!p) return;
    if (p->exec) hipGraphExecDestroy(p->exec);
    if (p->graph) hipGraphDestroy(p->graph);
    if (p->capture_stream) hipStreamDestroy(p->capture_stream);
    delete p;
}

extern "C" int lb_program_begin_record(void* prog) {
    LB_REQUIRE(prog != nullptr && g_recording == nullptr, "lb_program_begin_record: already recording");
    g_recording = (LbProgram*)prog;
    return 0;
}

extern "C" int lb_program_end_record(void* prog) {
    LB_REQUIRE(g_recording == (LbProgram*)prog, "lb_program_end_record: not recording this program");
    g_recording = nullptr;
    return 0;
}

extern "C" int lb_program_num_ops(void* prog) { return (int)((LbProgram*)prog)->ops.size(); }

extern "C" const char* lb_program_op_name(void* prog, int i) {
    LbProgram* p = (LbProgram*)prog;
    return (i >= 0 && i < (int)p->ops.size()) ? p->ops[i].name.c_str() : "";
}

extern "C" int lb_program_run_range(void* prog, int begin, int end, void* stream) {
    LbProgram* p = (LbProgram*)prog;
    LB_REQUIRE(p && begin >= 0 && end <= (int)p->ops.size() && begin <= end, "lb_program_run_range: range");
    for (int i = begin; i < end; ++i) {
        const int rc = p->ops[i].fn((hipStream_t)stream);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int lb_program_run(void* prog, void* stream) {
    return lb_program_run_range(prog, 0, lb_program_num_ops(prog), stream);
}

// Capture the whole launch sequence into a hipGraph (on a library-owned stream: nothing executes).
extern "C" int lb_program_instantiate(void* prog) {
    LbProgram* p = (LbProgram*)prog;
    LB_REQUIRE(p && !p->ops.empty(), "lb_program_instantiate: empty program");
    hipError_t e;
    if (!p->capture_stream) {
        e = hipStreamCreateWithFlags(&p->capture_stream, hipStreamNonBlocking);
        if (e != hipSuccess) { lb_set_error("lb_program_instantiate(stream)", e); return (int)e; }
    }
    if (p->exec) { hipGraphExecDestroy(p->exec); p->exec = nullptr; }
    if (p->graph) { hipGraphDestroy(p->graph); p->graph = nullptr; }
    e = hipStreamBeginCapture(p->capture_stream, hipStreamCaptureModeRelaxed);
    if (e != hipSuccess) { lb_set_error("lb_program_instantiate(begin)", e); return (int)e; }
    int rc = lb_program_run(prog, p->capture_stream);
    e = hipStreamEndCapture(p->capture_stream, &p->graph);
    if (rc) return rc;
    if (e != hipSuccess) { lb_set_error("lb_program_instantiate(end)", e); return (int)e; }
    e = hipGraphInstantiate(&p->exec, p->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) { lb_set_error("lb_program_instantiate(instantiate)", e); return (int)e; }
    return 0;
}

extern "C" int lb_program_launch(void* prog, void* stream) {
    LbProgram* p = (LbProgram*)prog;
    LB_REQUIRE(p != nullptr, "lb_program_launch: null program");
    if (!p->exec) return lb_program_run(prog, stream);
    hipError_t e = hipGraphLaunch(p->exec, (hipStream_t)stream);
    if (e != hipSuccess) { lb_set_error("lb_program_launch", e); return (int)e; }
    return 0;
}

// Plain device-to-device copy as a recordable op (refreshing program inputs inside a program).
static int copy_impl(void* dst, const void* src, long bytes, hipStream_t s) {
    hipError_t e = hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) { lb_set_error("lb_copy_d2d", e); return (int)e; }
    return 0;
}
extern "C" int lb_copy_d2d(void* dst, const void* src, long bytes, void* stream) {
    LB_REQUIRE(dst && src && bytes > 0, "lb_copy_d2d: arguments");
    LB_DISPATCH("lb_copy_d2d", copy_impl(dst, src, bytes, s));
}

// Per-op device time of one eager replay: N+1 hipEvents on `stream`, ms_out[i] = time from the
// event before op i to the event after it (kernel time + the launch boundary it really pays).
// Used by bench.py for the live roofline numbers; synchronises, so never call it while recording.
extern "C" int lb_program_time_ops(void* prog, void* stream_, float* ms_out) {
    LbProgram* p = (LbProgram*)prog;
    hipStream_t stream = (hipStream_t)stream_;
    LB_REQUIRE(p && ms_out && !lb_recording(), "lb_program_time_ops: arguments");
    const int n = (int)p->ops.size();
    std::vector<hipEvent_t> ev(n + 1);
    for (auto& e : ev) hipEventCreate(&e);
    hipEventRecord(ev[0], stream);
    int rc = 0;
    for (int i = 0; i < n && rc == 0; ++i) {
        rc = p->ops[i].fn(stream);
        hipEventRecord(ev[i + 1], stream);
    }
    hipError_t e = hipStreamSynchronize(stream);
    if (rc == 0 && e != hipSuccess) { lb_set_error("lb_program_time_ops", e); rc = (int)e; }
    if (rc == 0)
        for (int i = 0; i < n; ++i) hipEventElapsedTime(&ms_out[i], ev[i], ev[i + 1]);
    for (auto& x : ev) hipEventDestroy(x);
    return rc;
}
