"""Does partitioning the CUs between the generator and the front (encoder/FLAME/render) let them truly overlap?  (run on the GPU box)
CU-masked HIP streams (hipExtStreamCreateWithCUMask) wrapped as torch ExternalStreams."""
import ctypes, os, sys, tempfile, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from smirk_amd import masking
import synthdata as synth
from smirk_amd.pipeline import SmirkPipeline

hip = ctypes.CDLL("libamdhip64.so")
def masked_stream(lo, hi, total=256):
    words = (total + 31) // 32
    arr = (ctypes.c_uint32 * words)()
    for i in range(lo, hi):
        arr[i // 32] |= (1 << (i % 32))
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)

dev = torch.device("cuda:0")
sandbox = tempfile.mkdtemp()
enc, flame, rend, gen = bench.build_modules(sandbox, dev)
os.environ["SMIRK_ENCODER_SERIAL"] = "1"        # the encoder's own 3 streams are unmasked: keep it on the masked caller stream
cwd = os.getcwd(); os.chdir(sandbox); prob = masking.load_probabilities_per_FLAME_triangle().to(dev); os.chdir(cwd)
pipe = SmirkPipeline(enc, flame, rend, gen, prob)
B = 128
img = synth.synth_images(B, seed=0).to(dev)
hull = (synth.synth_generator_input(B, seed=1)[:, 3:4] != 0).float().to(dev)
masked = synth.synth_generator_input(B, seed=1)[:, 3:].contiguous().to(dev)
with torch.no_grad():
    out = pipe(img, hull_mask=hull)
rendered = out["rendered_img"]
torch.cuda.synchronize()

def t_stream(fn, stream, n=5):
    with torch.cuda.stream(stream), torch.no_grad():
        for _ in range(2): fn()
        stream.synchronize(); t = time.perf_counter()
        for _ in range(n): fn()
        stream.synchronize()
    return (time.perf_counter() - t) / n * 1e3

def front(): 
    e = enc(img); f = flame.forward(e); return rend.forward(f["vertices"], e["cam"])
def generate():
    return gen.forward_pair(rendered, masked)

full = torch.cuda.Stream()
print(f"unmasked: front {t_stream(front, full):.2f} ms, generator {t_stream(generate, full):.2f} ms")
for nfront in (32, 48, 64, 96):
    sg, sf = masked_stream(0, 256 - nfront), masked_stream(256 - nfront, 256)
    tg, tf = t_stream(generate, sg), t_stream(front, sf)
    # concurrent: both streams busy
    with torch.no_grad():
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5):
            with torch.cuda.stream(sg): generate()
            with torch.cuda.stream(sf): front()
        sg.synchronize(); sf.synchronize()
    tc = (time.perf_counter() - t) / 5 * 1e3
    print(f"front on {nfront} CUs / generator on {256 - nfront}: generator alone {tg:.2f} ms, front alone {tf:.2f} ms, both concurrently {tc:.2f} ms per batch")
# reference: both on ordinary streams
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
with torch.no_grad():
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5):
        with torch.cuda.stream(s1): generate()
        with torch.cuda.stream(s2): front()
    s1.synchronize(); s2.synchronize()
print(f"two ordinary streams, both concurrently: {(time.perf_counter() - t) / 5 * 1e3:.2f} ms per batch")
