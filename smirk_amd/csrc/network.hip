// Whole-network entries of libsmirk_hip.so: ONE host call enqueues every layer of a module's forward on the caller's stream.
//
//   smirk_generator_forward   SmirkGenerator.forward   (src/smirk_generator.py:51-86; _block :88-119; ResnetBlock :121-178)
//   smirk_backbone_forward    one sub-encoder of SmirkEncoder.forward: timm MobileNetV3-minimal features[-1] -> GAP -> Linear (+ clamps)
//                             (src/smirk_encoder.py:34-45, :66-73, :95-110; backbone per SURVEY.md App. A)
//
// These are pure schedules over the per-layer entries of conv.hip / conv_patch.hip / encoder_ops.hip / mbconv.hip: which kernel serves a
// layer is still decided by those dispatchers.  What moves here is the host-side walk (~60 launches for the generator, ~35-50 per
// backbone): from Python + ctypes (~15 ms per 128-frame step, more than the GPU time of the step) to straight C++ (a few hundred
// microseconds), and the activation memory from the framework's allocator to a caller-provided workspace that is laid out once:
//   * U-Net skip tensors e1..e4 have fixed slots,
//   * everything else rotates through three scratch slots sized for the largest temporary (a layer reads at most two temporaries —
//     input and residual — and writes one).
// No allocation, no synchronisation; every pointer is a device pointer except the weight structs (host structs of device pointers).
#include <sys/syscall.h>
#include <unistd.h>
#include <stdlib.h>

#include "common.h"

namespace {

struct Arena {
    char* base;
    size_t off;
    void* take(size_t bytes) {
        off = smirk_align_up(off, 256);
        void* p = base ? base + off : nullptr;
        off += bytes;
        return p;
    }
};

// three rotating scratch slots; pick() returns a slot that holds neither `a` nor `b`
struct Rot {
    void* slot[3];
    void* pick(const void* a, const void* b) const {
        for (int i = 0; i < 3; ++i)
            if (slot[i] != a && slot[i] != b) return slot[i];
        return nullptr;
    }
};

inline size_t act_bytes(int B, int H, int W, int C) { return (size_t)B * H * W * C * 4; }

struct GenPlan {
    void* x;         // packed network input [B][H][W][cin_pad]
    void* e[4];      // skip tensors
    Rot rot;
    size_t total;
};

GenPlan plan_generator(const SmirkGeneratorWeights* w, int B, int H, int W, void* ws) {
    Arena a{(char*)ws, 0};
    GenPlan p;
    const int f = w->features;
    p.x = a.take(act_bytes(B, H, W, w->cin_pad));
    for (int l = 0; l < 4; ++l) p.e[l] = a.take(act_bytes(B, H >> l, W >> l, f << l));
    const size_t big = act_bytes(B, H, W, f);                       // the largest temporary: enc1conv1 out, up1, dec1conv1 out
    for (int i = 0; i < 3; ++i) p.rot.slot[i] = a.take(big);
    p.total = smirk_align_up(a.off, 256);
    return p;
}

int conv_call(bool split, const SmirkConvDesc& d, const void* in0, const void* in1, const SmirkConvLayer& L, const void* residual, void* out,
              void* stream) {
    return split ? smirk_conv_igemm_f16x3(&d, in0, in1, L.w, L.scale, L.shift, residual, out, stream)
                 : smirk_conv_igemm_f32(&d, (const float*)in0, (const float*)in1, (const float*)L.w, L.scale, L.shift, (const float*)residual,
                                        (float*)out, stream);
}

SmirkConvDesc desc3x3(int B, int H, int W, int c0, int c1, int cout, bool reflect, bool relu) {
    SmirkConvDesc d;
    d.B = B; d.H = H; d.W = W; d.C0 = c0; d.C1 = c1; d.Cout = cout; d.KH = d.KW = 3; d.stride = 1; d.pad_t = d.pad_l = 1;
    d.Ho = H; d.Wo = W; d.pad_mode = reflect ? SMIRK_PAD_REFLECT : SMIRK_PAD_ZERO; d.act = relu ? SMIRK_ACT_RELU : SMIRK_ACT_NONE;
    d.out_mode = SMIRK_OUT_NHWC;
    return d;
}

SmirkConvDesc desc1x1(int B, int H, int W, int cin, int cout, bool relu, bool convt) {
    SmirkConvDesc d;
    d.B = B; d.H = H; d.W = W; d.C0 = cin; d.C1 = 0; d.Cout = cout; d.KH = d.KW = 1; d.stride = 1; d.pad_t = d.pad_l = 0;
    d.Ho = H; d.Wo = W; d.pad_mode = SMIRK_PAD_ZERO; d.act = relu ? SMIRK_ACT_RELU : SMIRK_ACT_NONE;
    d.out_mode = convt ? SMIRK_OUT_CONVT2X2 : SMIRK_OUT_NHWC;
    return d;
}

#define TRY(expr)                      \
    do {                               \
        const int rc_ = (expr);        \
        if (rc_ != SMIRK_OK) return rc_; \
    } while (0)

bool gen_args_ok(const SmirkGeneratorWeights* w, int B, int H, int W) {
    if (!w || B <= 0 || H <= 0 || W <= 0 || H % 16 || W % 16) return false;
    if (w->res_blocks < 0 || w->res_blocks > SMIRK_GEN_MAX_RES || w->features <= 0 || w->out_channels <= 0 || w->out_channels > 4) return false;
    if (w->precision == SMIRK_PRECISION_F16X3) return w->features % 8 == 0 && w->cin_pad == 8 && w->in_channels <= 8;
    return w->precision == SMIRK_PRECISION_F32 && w->features % 8 == 0 && w->cin_pad % 4 == 0 && w->cin_pad >= w->in_channels;
}

}  // namespace

extern "C" size_t smirk_generator_workspace_bytes(const SmirkGeneratorWeights* w, int B, int H, int W) {
    if (!gen_args_ok(w, B, H, W)) return 0;
    return plan_generator(w, B, H, W, nullptr).total;
}

extern "C" int smirk_generator_forward(const SmirkGeneratorWeights* w, const float* a, int Ca, const float* b, int Cb, float* y, int B, int H,
                                       int W, void* const* taps, void* ws, size_t ws_bytes, void* stream) {
    if (!gen_args_ok(w, B, H, W) || !a || !y || !ws || Ca <= 0 || Cb < 0 || Ca + Cb != w->in_channels || (Cb > 0 && !b)) return SMIRK_ERR_BAD_ARG;
    if (!w->final_w || !w->final_b) return SMIRK_ERR_BAD_ARG;
    const GenPlan p = plan_generator(w, B, H, W, ws);
    if (ws_bytes < p.total) return SMIRK_ERR_WORKSPACE;
    const bool split = w->precision == SMIRK_PRECISION_F16X3;
    const int f = w->features;

    // ---- cat + NCHW -> NHWC (+ split) of the network input (smirk_trainer.py:94, demo.py:167) ----------------------------------------
    if (split) {
        TRY(smirk_pack_generator_input_split16(a, Ca, b, Cb, p.x, B, H, W, stream));
    } else if (Cb == 0) {
        TRY(smirk_nchw_to_nhwc_pad(a, (float*)p.x, B, Ca, H, W, w->cin_pad, stream));
    } else {
        if (Ca != 3 || Cb != 3 || w->cin_pad != 8) return SMIRK_ERR_UNSUPPORTED;
        TRY(smirk_pack_generator_input(a, b, (float*)p.x, B, H, W, stream));
    }
    auto pool = [&](const void* in, void* out, int h, int wd, int c) {
        return split ? smirk_maxpool2x2_split16(in, out, B, h, wd, c, stream) : smirk_maxpool2x2_nhwc((const float*)in, (float*)out, B, h, wd, c, stream);
    };
    auto tap = [&](int i, const void* src, size_t bytes) {
        if (taps && taps[i]) (void)hipMemcpyAsync(taps[i], src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream);
    };

    // ---- how the deep part of the network is launched (decided first: it sets where the whole-batch launches stop) ---------------------------------------------
    // The layers at H/4 and below rarely fill whole rounds of workgroups on the 256 CUs (14 x 14 x 512: 196 B / 256 tiles x 4 = 1.53 rounds at 128 frames = 2 rounds
    // of time, 3.06 -> 4 at 256; the 28 x 28 and 56 x 56 layers likewise), and each waits for the one before it.  Frames are independent, so this SECTION — encoder3,
    // encoder4, bottleneck, ResNet blocks, decoder4, decoder3 — runs as TWO (optionally three) sub-batch chains on as many streams (the caller's and library-owned
    // side streams, forked / joined with events): while one chain's layer drains its partial last round the other chain's workgroups take the free CUs.
    // Bit-identical results (batch invariance, tests/test_scale_gpu.py run with it forced).  Measured with the RCCL gather enqueued, same box (profiles/r04l_, r04m_,
    // r04n_): +3.2 % at 128 frames per pass, +4.5 % at 256, +1.2 % at 1024 for the H/8 + H/16 part -> taken when > 5 % of a 14 x 14 layer's last round would idle.
    // $SMIRK_GEN_SPLIT_CHAINS=0 / 2 / 3 selects the number of chains (the A/B switch of tests/test_generator_gpu.py).  The section starts at H/8: starting it at H/4
    // measured 8,801 vs 8,857 faces/s at 128 frames, 9,406 vs 9,473 at 256, 9,872 vs 9,782 at 1024 (profiles/r04n_chain_from.txt; the 56 x 56 layers fill their rounds) and
    // the switch for it left the library.  Batches below 16 frames are never split by the heuristic (a fork / join costs more than the idle part of their only round).
    const int h16 = H >> 4, w16 = W >> 4, c16 = f << 4;
    const int L0 = 3;
    int nchain = 1;
    if (B >= 2 && !taps && !g_smirk_prof_on) {                      // (taps copy whole tensors; the launch profiler times launches on ONE stream)
        static int n_cu = 0;
        if (n_cu == 0) {
            int dev = 0, cus = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) { (void)hipGetLastError(); cus = 256; }
            n_cu = cus;
        }
        const double rounds = ((double)B * h16 * w16 / 256.0) * (c16 / 128.0) / n_cu;      // 256 x 128 tiles of a 14 x 14 layer per CU
        const double full = (double)(long long)(rounds + 0.999999);
        if (B >= 16 && full > 0 && (full - rounds) / full > 0.05) nchain = 2;
        if (const char* e = getenv("SMIRK_GEN_SPLIT_CHAINS")) nchain = e[0] == '0' ? 1 : e[0] == '3' ? 3 : 2;
        if (nchain > B) nchain = B;
    }
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (nchain > 1 && (hipStreamIsCapturing((hipStream_t)stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone)) { (void)hipGetLastError(); nchain = 1; }
    // Side streams and the fork / join events are owned per HOST THREAD and device (thread_local): two threads driving the same device neither race on the lazy
    // creation nor record / wait on each other's events; created once, reused by every forward of that thread (re-recording an event only affects later waits).
    hipStream_t side[2] = {nullptr, nullptr};
    hipEvent_t fork = nullptr, join[2] = {nullptr, nullptr};
    if (nchain > 1) {
        // RAII: a host thread that ends releases its side streams and events (a thread pool of short-lived workers leaked 2 streams + 3 events per thread; advisor r05).
        // The destructor runs at thread exit; work still queued on a side stream completes first (hipStreamDestroy defers the release until the stream drains).
        struct ChainResources {
            hipStream_t side[64][2] = {};
            hipEvent_t ev[64][3] = {};
            ~ChainResources() {
                // the MAIN thread's thread_locals are destroyed inside exit(), where the HIP runtime may already be shutting down: the process exit reclaims them
                if ((long)syscall(SYS_gettid) == (long)getpid()) return;
                int cur = 0;
                const bool have = hipGetDevice(&cur) == hipSuccess;
                for (int d = 0; d < 64; ++d) {
                    bool any = false;
                    for (int k = 0; k < 2; ++k) any = any || side[d][k];
                    for (int k = 0; k < 3; ++k) any = any || ev[d][k];
                    if (!any) continue;
                    if (hipSetDevice(d) != hipSuccess) { (void)hipGetLastError(); continue; }
                    for (int k = 0; k < 2; ++k) if (side[d][k]) (void)hipStreamDestroy(side[d][k]);
                    for (int k = 0; k < 3; ++k) if (ev[d][k]) (void)hipEventDestroy(ev[d][k]);
                }
                if (have) (void)hipSetDevice(cur);
                (void)hipGetLastError();
            }
        };
        thread_local ChainResources chain_res;
        hipStream_t (&side_dev)[64][2] = chain_res.side;
        hipEvent_t (&ev_dev)[64][3] = chain_res.ev;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); nchain = 1; }
        else {
            for (int k = 0; k < 3 && nchain > 1; ++k)
                if (!ev_dev[dev][k] && hipEventCreateWithFlags(&ev_dev[dev][k], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); ev_dev[dev][k] = nullptr; nchain = 1; }
            for (int k = 0; k + 1 < nchain; ++k) {
                if (!side_dev[dev][k] && hipStreamCreateWithFlags(&side_dev[dev][k], hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); side_dev[dev][k] = nullptr; nchain = 1; break; }
                side[k] = side_dev[dev][k];
            }
            fork = ev_dev[dev][0]; join[0] = ev_dev[dev][1]; join[1] = ev_dev[dev][2];
        }
    }
    // ---- encoder levels above the section, whole batch (smirk_generator.py:52-57) --------------------------------------------------------------------------
    const void* cur = p.x;
    int cin = w->cin_pad;
    for (int l = 0; l < L0; ++l) {
        const int h = H >> l, wd = W >> l, c = f << l;
        if (l == 0 && split && smirk_enc1_fused_supported(cin, f, h, wd)) {
            // encoder1 + pool1 in one launch (enc1_fused.hip): the 32-channel full-resolution tensor between the two convolutions never reaches HBM
            void* pl = p.rot.pick(nullptr, nullptr);
            TRY(smirk_enc1_fused_split16(cur, w->enc[0][0].w, w->enc[0][0].scale, w->enc[0][0].shift, w->enc[0][1].w, w->enc[0][1].scale,
                                         w->enc[0][1].shift, p.e[0], pl, B, h, wd, stream));
            tap(0, p.e[0], act_bytes(B, h, wd, c));
            cur = pl;
            cin = c;
            continue;
        }
        void* t1 = p.rot.pick(cur, nullptr);
        TRY(conv_call(split, desc3x3(B, h, wd, cin, 0, c, false, true), cur, nullptr, w->enc[l][0], nullptr, t1, stream));
        void* t2 = p.e[l];
        if (split) {                                       // conv2 + BN + ReLU + pool in one launch where a kernel with a fused pool serves the shape
            void* pl = p.rot.pick(t1, nullptr);
            const SmirkConvDesc d2 = desc3x3(B, h, wd, c, 0, c, false, true);
            const int rc = smirk_conv3x3_pool_f16x3(&d2, t1, nullptr, w->enc[l][1].w, w->enc[l][1].scale, w->enc[l][1].shift, t2, pl, stream);
            if (rc == SMIRK_OK) {
                tap(l, t2, act_bytes(B, h, wd, c));
                cur = pl;
                cin = c;
                continue;
            }
            if (rc != SMIRK_ERR_UNSUPPORTED) return rc;
        }
        TRY(conv_call(split, desc3x3(B, h, wd, c, 0, c, false, true), t1, nullptr, w->enc[l][1], nullptr, t2, stream));
        tap(l, t2, act_bytes(B, h, wd, c));
        void* pl = p.rot.pick(nullptr, nullptr);
        TRY(pool(t2, pl, h, wd, c));
        cur = pl;
        cin = c;
    }
    // ---- the section (smirk_generator.py:56-68, :121-178) -----------------------------------------------------------------------------------------------
    // Memory: temporaries rotate through the three scratch slots, each sized for the largest tensor of the network (B x H x W x f elements = 4x the largest tensor of
    // this section).  The chains run at different paces and would otherwise put tensors of DIFFERENT shapes (so different frame offsets) into one slot at the same
    // time: chain k keeps its temporaries in QUARTER k + 1 of every slot, indexed from frame 0; only the tensors that cross the section's boundary use the whole-batch
    // layout — its input (the pooled tensor of the level above), the skip tensors, its output — which all lie inside quarter 0 with frame ranges that are disjoint
    // between the chains; the output never shares a slot with the input (one chain may finish while another still reads its input frames).
    const size_t quarter = (act_bytes(B, H, W, f) / 4) & ~(size_t)255;
    auto poolh = [&](const void* in, void* out, int nb, int h, int wd, int c, void* strm) {
        return split ? smirk_maxpool2x2_split16(in, out, nb, h, wd, c, strm) : smirk_maxpool2x2_nhwc((const float*)in, (float*)out, nb, h, wd, c, strm);
    };
    auto deep = [&](int chain, int b0, int nb, void* strm, const void* in, const void*& out) -> int {
        const size_t priv = nchain > 1 ? quarter * (size_t)(chain + 1) : 0;
        auto N = [&](const void* base, int h, int wd, int c) { return (const void*)((const char*)base + act_bytes(b0, h, wd, c)); };    // whole-batch layout
        auto Nw = [&](void* base, int h, int wd, int c) { return (void*)((char*)base + act_bytes(b0, h, wd, c)); };
        auto P = [&](const void* base) { return (const void*)((const char*)base + priv); };                                              // chain-private
        auto Pw = [&](void* base) { return (void*)((char*)base + priv); };
        const void* cur = in;
        bool cur_nat = true;
        int ci = f << (L0 - 1);
        for (int l = L0; l < 4; ++l) {                              // encoder levels of the section: conv, conv -> skip tensor, pool
            const int h = H >> l, wd = W >> l, c = f << l;
            void* t1 = p.rot.pick(cur, nullptr);
            TRY(conv_call(split, desc3x3(nb, h, wd, ci, 0, c, false, true), cur_nat ? N(cur, h, wd, ci) : P(cur), nullptr, w->enc[l][0], nullptr, Pw(t1), strm));
            TRY(conv_call(split, desc3x3(nb, h, wd, c, 0, c, false, true), P(t1), nullptr, w->enc[l][1], nullptr, Nw(p.e[l], h, wd, c), strm));
            tap(l, p.e[l], act_bytes(B, h, wd, c));                 // (taps => one chain over the whole batch on the caller's stream)
            void* pl = p.rot.pick(nullptr, nullptr);
            TRY(poolh(N(p.e[l], h, wd, c), Pw(pl), nb, h, wd, c, strm));
            cur = pl; cur_nat = false; ci = c;
        }
        const int c8 = f << 3;
        void* b1 = p.rot.pick(cur, nullptr);                        // bottleneck
        TRY(conv_call(split, desc3x3(nb, h16, w16, c8, 0, c16, false, true), P(cur), nullptr, w->enc[4][0], nullptr, Pw(b1), strm));
        void* b2 = p.rot.pick(b1, nullptr);
        TRY(conv_call(split, desc3x3(nb, h16, w16, c16, 0, c16, false, true), P(b1), nullptr, w->enc[4][1], nullptr, Pw(b2), strm));
        tap(4, b2, act_bytes(B, h16, w16, c16));
        const void* cur16 = b2;                                     // ResNet blocks: reflect pad, conv-BN-ReLU, reflect pad, conv-BN, + x
        for (int k = 0; k < w->res_blocks; ++k) {
            void* t = p.rot.pick(cur16, nullptr);
            TRY(conv_call(split, desc3x3(nb, h16, w16, c16, 0, c16, true, true), P(cur16), nullptr, w->res[k][0], nullptr, Pw(t), strm));
            void* o = p.rot.pick(cur16, t);
            TRY(conv_call(split, desc3x3(nb, h16, w16, c16, 0, c16, true, false), P(t), nullptr, w->res[k][1], P(cur16), Pw(o), strm));
            cur16 = o;
        }
        tap(5, cur16, act_bytes(B, h16, w16, c16));
        const void* dc = cur16;
        for (int l = 3; l >= L0; --l) {                             // decoder levels of the section: ConvTranspose2d, two-source conv with the skip, conv
            const int h = H >> l, wd = W >> l, c = f << l;
            void* up = p.rot.pick(dc, nullptr);
            TRY(conv_call(split, desc1x1(nb, h / 2, wd / 2, 2 * c, c, false, true), P(dc), nullptr, w->up[3 - l], nullptr, Pw(up), strm));
            void* d1 = p.rot.pick(up, nullptr);
            TRY(conv_call(split, desc3x3(nb, h, wd, c, c, c, false, true), P(up), N(p.e[l], h, wd, c), w->dec[3 - l][0], nullptr, Pw(d1), strm));
            const bool last = l == L0;
            void* d2 = last ? p.rot.pick(d1, in) : p.rot.pick(d1, nullptr);
            TRY(conv_call(split, desc3x3(nb, h, wd, c, 0, c, false, true), P(d1), nullptr, w->dec[3 - l][1], nullptr, last ? Nw(d2, h, wd, c) : Pw(d2), strm));
            tap(6 + (3 - l), d2, act_bytes(B, h, wd, c));
            dc = d2;
        }
        out = dc;
        return SMIRK_OK;
    };
    const void* dcur = nullptr;
    if (nchain > 1) {
        int rc = hipEventRecord(fork, (hipStream_t)stream) == hipSuccess ? SMIRK_OK : SMIRK_ERR_LAUNCH;
        const void* last = nullptr;
        bool forked[2] = {false, false};
        for (int k = nchain - 1; rc == SMIRK_OK && k >= 0; --k) {   // chain k owns frames [B k / n, B (k+1) / n); chain 0 runs on the caller's stream, enqueued last
            const int b0 = (int)((long long)B * k / nchain), b1 = (int)((long long)B * (k + 1) / nchain);
            hipStream_t st_k = k == 0 ? (hipStream_t)stream : side[k - 1];
            if (k > 0 && hipStreamWaitEvent(st_k, fork, 0) != hipSuccess) { rc = SMIRK_ERR_LAUNCH; break; }
            if (k > 0) forked[k - 1] = true;
            rc = deep(k, b0, b1 - b0, (void*)st_k, cur, last);
            if (rc == SMIRK_OK && k > 0 && hipEventRecord(join[k - 1], st_k) != hipSuccess) rc = SMIRK_ERR_LAUNCH;
        }
        for (int k = 0; rc == SMIRK_OK && k + 1 < nchain; ++k)
            if (hipStreamWaitEvent((hipStream_t)stream, join[k], 0) != hipSuccess) rc = SMIRK_ERR_LAUNCH;
        if (rc != SMIRK_OK) {
            // a chain failed part-way: whatever was enqueued on the side streams still reads / writes the caller's workspace.  The caller's stream must not run
            // ahead of it (the workspace may be reused or freed in stream order), so the side streams are drained before the error is returned.
            for (int k = 0; k < 2; ++k)
                if (forked[k]) (void)hipStreamSynchronize(side[k]);
            return rc;
        }
        dcur = last;
    } else {
        TRY(deep(0, 0, B, stream, cur, dcur));
    }
    // ---- decoder4..1 (smirk_generator.py:65-75): ConvTranspose2d k2 s2, cat with the skip (two-source conv), double conv -------------------
    for (int l = L0 - 1; l >= 0; --l) {                              // decoder levels above the section, whole batch
        const int h = H >> l, wd = W >> l, c = f << l;              // output resolution of this level
        void* up = p.rot.pick(dcur, nullptr);
        TRY(conv_call(split, desc1x1(B, h / 2, wd / 2, 2 * c, c, false, true), dcur, nullptr, w->up[3 - l], nullptr, up, stream));
        void* t1 = p.rot.pick(up, nullptr);
        TRY(conv_call(split, desc3x3(B, h, wd, c, c, c, false, true), up, p.e[l], w->dec[3 - l][0], nullptr, t1, stream));
        if (l == 0 && split && !taps && f == 32 && H >= 64) {
            // network tail in one launch: dec1conv2 + BN + ReLU + final 1x1 conv + sigmoid (dec1 never reaches HBM)
            SmirkConvDesc d = desc3x3(B, h, wd, c, 0, c, false, true);
            const int rc = smirk_conv3x3_tail_f16x3(&d, t1, nullptr, w->dec[3][1].w, w->dec[3][1].scale, w->dec[3][1].shift, w->final_w, w->final_b,
                                                    y, w->out_channels, stream);
            if (rc != SMIRK_ERR_UNSUPPORTED) return rc;
        }
        void* t2 = p.rot.pick(t1, nullptr);
        TRY(conv_call(split, desc3x3(B, h, wd, c, 0, c, false, true), t1, nullptr, w->dec[3 - l][1], nullptr, t2, stream));
        tap(6 + (3 - l), t2, act_bytes(B, h, wd, c));
        dcur = t2;
    }
    return split ? smirk_conv1x1_sigmoid_nchw_split16(dcur, w->final_w, w->final_b, y, B, H, W, f, w->out_channels, stream)
                 : smirk_conv1x1_sigmoid_nchw((const float*)dcur, w->final_w, w->final_b, y, B, H, W, f, w->out_channels, stream);
}

// ------------------------------------------------------------------------------------------------------------------------------------
// MobileNetV3-minimal backbone + pooled linear head
// ------------------------------------------------------------------------------------------------------------------------------------
namespace {

bool backbone_args_ok(const SmirkBackboneWeights* w, int B, int H, int W) {
    if (!w || B <= 0 || H < 32 || W < 32 || w->n_blocks <= 0 || w->n_blocks > SMIRK_BACKBONE_MAX_BLOCKS || w->n_out < 0) return false;
    if (w->precision != SMIRK_PRECISION_F16X3 && w->precision != SMIRK_PRECISION_F32) return false;
    if (w->n_out > 0 && (!w->head_w || !w->head_b)) return false;                    // n_out == 0: features only, no head
    return w->stem.w && w->stem.scale && w->stem.shift && w->stem_cout > 0;
}

// largest activation (in elements) any layer of the backbone reads or writes, walking the block list
size_t backbone_max_elems(const SmirkBackboneWeights* w, int B, int H, int W) {
    int h = (H + 1) / 2, wd = (W + 1) / 2;
    size_t m = (size_t)B * h * wd * w->stem_cout;
    for (int i = 0; i < w->n_blocks; ++i) {
        const SmirkMbBlock& b = w->blocks[i];
        const int ho = (h + b.stride - 1) / b.stride, wo = (wd + b.stride - 1) / b.stride;
        const size_t in_mid = (size_t)B * h * wd * b.mid, out_mid = (size_t)B * ho * wo * b.mid, out = (size_t)B * ho * wo * b.cout;
        if (b.kind == 1 && in_mid > m) m = in_mid;
        if (b.kind != 2 && out_mid > m) m = out_mid;
        if (out > m) m = out;
        h = ho; wd = wo;
    }
    return m;
}

struct BackbonePlan {
    Rot rot;
    float* pooled;
    size_t total;
};

BackbonePlan plan_backbone(const SmirkBackboneWeights* w, int B, int H, int W, void* ws) {
    Arena a{(char*)ws, 0};
    BackbonePlan p;
    const size_t big = backbone_max_elems(w, B, H, W) * 4;
    for (int i = 0; i < 3; ++i) p.rot.slot[i] = a.take(big);
    p.pooled = (float*)a.take((size_t)B * w->feat_ch * 4);
    p.total = smirk_align_up(a.off, 256);
    return p;
}

}  // namespace

extern "C" size_t smirk_backbone_workspace_bytes(const SmirkBackboneWeights* w, int B, int H, int W) {
    if (!backbone_args_ok(w, B, H, W)) return 0;
    return plan_backbone(w, B, H, W, nullptr).total;
}

extern "C" int smirk_backbone_forward(const SmirkBackboneWeights* w, const float* img, int B, int H, int W, float* out, void* feat_out,
                                      void* ws, size_t ws_bytes, void* stream) {
    if (!backbone_args_ok(w, B, H, W) || !img || !ws || (w->n_out > 0 && !out) || (w->n_out == 0 && !feat_out)) return SMIRK_ERR_BAD_ARG;
    const BackbonePlan p = plan_backbone(w, B, H, W, ws);
    if (ws_bytes < p.total) return SMIRK_ERR_WORKSPACE;
    const bool split = w->precision == SMIRK_PRECISION_F16X3;
    const bool no_image = getenv("SMIRK_DISABLE_MBCONV_IMAGE") != nullptr;                                                       // A/B switch (tests)
    const bool no_tile = getenv("SMIRK_DISABLE_MBCONV_TILE") != nullptr;      // A/B switch (tests): the 24-48-channel blocks stay on mbconv_fused_kernel
    const bool no_fuse = getenv("SMIRK_DISABLE_MBCONV_FUSED") != nullptr;                                                         // A/B switch (tests)
    const bool fuse_ds = false;       // DepthwiseSeparable blocks stay unfused (no expanded tensor to save: 202 vs 200 us at stride 1, 97 vs 66 at stride 2)
    int h = (H + 1) / 2, wd = (W + 1) / 2;
    void* x = p.rot.slot[0];
    // stem + first DepthwiseSeparable block in one launch (encoder_head.hip): the 16-channel 112 x 112 tensors between them never reach HBM
    int first_block = 0;
    {
        const SmirkMbBlock& b0 = w->blocks[0];
        const bool no_head = getenv("SMIRK_DISABLE_ENCODER_HEAD_FUSED") != nullptr;                                               // A/B switch (tests)
        if (split && !no_head && smirk_encoder_head_supported(w->stem_cout, b0.kind, b0.cin, b0.mid, b0.cout, b0.stride, b0.skip)) {
            TRY(smirk_encoder_head_fused_split16(img, (const float*)w->stem.w, w->stem.scale, w->stem.shift, (const float*)b0.dw.w, b0.dw.scale, b0.dw.shift,
                                                 b0.pw.w, b0.pw.scale, b0.pw.shift, b0.skip ? 1 : 0, x, B, H, W, b0.stride, stream));
            h = (h + b0.stride - 1) / b0.stride; wd = (wd + b0.stride - 1) / b0.stride;
            first_block = 1;
        }
    }
    if (!first_block)
    TRY(split ? smirk_stem_conv_s2_split16(img, (const float*)w->stem.w, w->stem.scale, w->stem.shift, x, B, H, W, w->stem_cout, stream)
              : smirk_stem_conv_s2(img, (const float*)w->stem.w, w->stem.scale, w->stem.shift, (float*)x, B, H, W, w->stem_cout, stream));
    auto pointwise = [&](const void* in, int hh, int ww, int cin, int cout, const SmirkConvLayer& L, bool relu, const void* res, void* o) {
        return conv_call(split, desc1x1(B, hh, ww, cin, cout, relu, false), in, nullptr, L, res, o, stream);
    };
    auto depthwise = [&](const void* in, int hh, int ww, int c, int stride, const SmirkConvLayer& L, void* o) {
        return split ? smirk_dwconv3x3_split16(in, (const float*)L.w, L.scale, L.shift, o, B, hh, ww, c, stride, 1, stream)
                     : smirk_dwconv3x3((const float*)in, (const float*)L.w, L.scale, L.shift, (float*)o, B, hh, ww, c, stride, 1, stream);
    };
    int c = first_block ? w->blocks[0].cout : w->stem_cout;
    for (int i = first_block; i < w->n_blocks; ++i) {
        const SmirkMbBlock& b = w->blocks[i];
        if (b.cin != c) return SMIRK_ERR_BAD_ARG;
        const int ho = (h + b.stride - 1) / b.stride, wo = (wd + b.stride - 1) / b.stride;
        const void* res = b.skip ? x : nullptr;
        if (b.kind == 2) {                                          // ConvBnAct 1x1
            void* o = p.rot.pick(x, nullptr);
            TRY(pointwise(x, h, wd, b.cin, b.cout, b.pw, true, nullptr, o));
            x = o;
        } else if (split && !no_fuse && !no_image && !no_tile && b.kind == 1 && b.cin <= 48 && smirk_mbconv_image_supported(h, wd, b.cin, b.mid, b.cout, b.stride)) {
            // stride-1 blocks of the 56 x 56 / 28 x 28 stages (14 x 14 halo tiles) and of the small backbone's 14 x 14 stage (whole images), 24-48 channels:
            // the image-resident kernel instead of the 8 x 8-tile kernel below (round 5)
            void* o = p.rot.pick(x, nullptr);
            TRY(smirk_mbconv_image_split16(x, b.pw.w, b.pw.scale, b.pw.shift, (const float*)b.dw.w, b.dw.scale, b.dw.shift, b.pwl.w, b.pwl.scale, b.pwl.shift,
                                           b.skip ? 1 : 0, o, B, h, wd, b.cin, b.mid, b.cout, stream));
            x = o;
        } else if (split && !no_fuse && (b.kind == 1 || fuse_ds) && smirk_mbconv_supported(b.cin, b.mid, b.cout, b.stride)) {
            void* o = p.rot.pick(x, nullptr);
            const SmirkConvLayer& proj = b.kind == 1 ? b.pwl : b.pw;
            TRY(smirk_mbconv_fused_split16(x, b.kind == 1 ? b.pw.w : nullptr, b.kind == 1 ? b.pw.scale : nullptr, b.kind == 1 ? b.pw.shift : nullptr,
                                           (const float*)b.dw.w, b.dw.scale, b.dw.shift, proj.w, proj.scale, proj.shift, b.skip ? 1 : 0, o, B, h, wd,
                                           b.cin, b.mid, b.cout, b.stride, stream));
            x = o;
        } else if (split && !no_fuse && !no_image && b.kind == 1 && smirk_mbconv_image_supported(h, wd, b.cin, b.mid, b.cout, b.stride)) {
            void* o = p.rot.pick(x, nullptr);                       // <= 14 x 14, 80-112 channels: whole images per workgroup (mbconv_image.hip)
            TRY(smirk_mbconv_image_split16(x, b.pw.w, b.pw.scale, b.pw.shift, (const float*)b.dw.w, b.dw.scale, b.dw.shift, b.pwl.w, b.pwl.scale, b.pwl.shift,
                                           b.skip ? 1 : 0, o, B, h, wd, b.cin, b.mid, b.cout, stream));
            x = o;
        } else if (b.kind == 0) {                                   // DepthwiseSeparable: dw + BN + ReLU -> pw + BN (+ x)
            void* t = p.rot.pick(x, nullptr);
            TRY(depthwise(x, h, wd, b.cin, b.stride, b.dw, t));
            void* o = p.rot.pick(x, t);
            TRY(pointwise(t, ho, wo, b.cin, b.cout, b.pw, false, res, o));
            x = o;
        } else {                                                    // InvertedResidual: pw + BN + ReLU -> dw + BN + ReLU -> pwl + BN (+ x)
            void* t = p.rot.pick(x, nullptr);
            TRY(pointwise(x, h, wd, b.cin, b.mid, b.pw, true, nullptr, t));
            void* u = p.rot.pick(x, t);
            TRY(depthwise(t, h, wd, b.mid, b.stride, b.dw, u));
            void* o = p.rot.pick(x, u);
            TRY(pointwise(u, ho, wo, b.mid, b.cout, b.pwl, false, res, o));
            x = o;
        }
        h = ho; wd = wo; c = b.cout;
    }
    if (c != w->feat_ch) return SMIRK_ERR_BAD_ARG;
    if (feat_out) (void)hipMemcpyAsync(feat_out, x, act_bytes(B, h, wd, c), hipMemcpyDeviceToDevice, (hipStream_t)stream);
    if (w->n_out == 0) return SMIRK_OK;
    TRY(split ? smirk_gap_linear_split16(x, w->head_w, w->head_b, out, p.pooled, B, h * wd, c, w->n_out, stream)
              : smirk_gap_linear((const float*)x, w->head_w, w->head_b, out, p.pooled, B, h * wd, c, w->n_out, stream));
    if (w->clamp_n_exp >= 0) TRY(smirk_expression_clamps(out, B, w->clamp_n_exp, stream));
    return SMIRK_OK;
}
