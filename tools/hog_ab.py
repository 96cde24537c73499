#!/usr/bin/env python3
"""How does the training step tolerate a collective's long-running workgroups?  A hog kernel (tools/hog.hip: N workgroups
spinning for ~T us, launched on its own stream a few times per step, like the gradient buckets' all-reduce) runs beside the
step, with the persistent GEMM kernels (static tile ownership: a workgroup that cannot start delays its tiles) and without.
usage: python tools/hog_ab.py [batch]"""
import ctypes, os, subprocess, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pero_pretraining_amd import _lib, functional as F

here = os.path.dirname(os.path.abspath(__file__))
so = "/tmp/hog.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "--offload-arch=gfx950", "-shared", "-fPIC", os.path.join(here, "hog.hip"), "-o", so])
hog = ctypes.CDLL(so)
hog.hog_launch.argtypes = [ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda", 0)
model, opt, sched, trainer = bench.build(dev, True)
batches = bench.synthetic(0, B, dev)
sink = torch.zeros(1, dtype=torch.int32, device=dev)
hstream = torch.cuda.Stream()
TICKS_PER_US = 100  # wall_clock64 runs at 100 MHz

def run(n, blocks, us, per_step):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        sched.update_learning_rate(i)
        if blocks:
            # the hogs are spread over the step by the host's enqueue order only: launch them up front on their own stream
            for _ in range(per_step):
                hog.hog_launch(blocks, us * TICKS_PER_US, sink.data_ptr(), hstream.cuda_stream)
        trainer.train_step_prepared(*batches[i % 2])
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

run(2, 0, 0, 0)
for pers in (1, 0):
    _lib.lib().pero_set_option(b"gemm_persistent", pers)
    for blocks, us, per_step in ((0, 0, 0), (32, 300, 7), (64, 300, 7), (32, 1000, 7)):
        run(1, blocks, us, per_step)
        t = min(run(3, blocks, us, per_step), run(3, blocks, us, per_step))
        print(f"gemm_persistent={pers} hog {blocks:3d} workgroups x {us:4d} us x {per_step}/step: {t:.2f} ms/step")
_lib.lib().pero_set_option(b"gemm_persistent", 1)
