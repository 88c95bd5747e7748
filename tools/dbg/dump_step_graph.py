import os, sys, ctypes, torch
sys.path.insert(0, os.getcwd())
import relationnetworks_clevr_amd as pkg
from relationnetworks_clevr_amd import dp
orig = dp.DataParallelTrainer._capture
def cap(self, img, qst, label):
    self._static = (img.clone(), qst.clone(), label.clone())
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2): self._fwd_bwd(*self._static)
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(graph):
        self._loss = self._fwd_bwd(*self._static)
        if self._opt_in_graph: self._fused_opt.step_dev()
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipGraphDebugDotPrint.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint]
    print("dot rc", hip.hipGraphDebugDotPrint(graph.raw_cuda_graph(), b"gpurun_out/step_graph.dot", 1))
    graph.instantiate()
    self._graph = graph
dp.DataParallelTrainer._capture = cap
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-other-modes", "--no-parity", "--no-kernel-timing", "--steps", "3", "--warmup", "1"]
import runpy; runpy.run_path("bench.py", run_name="__main__")
