import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from openeat_amd import ddp, hip
hip.GEMM_PRECISION = 3
rank, local, world = ddp.init_from_env()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
from openeat_amd.engine import TrainEngine
from openeat_amd.models.asr_model import ASRModel
import bench
torch.manual_seed(777)
model = ASRModel(80, bench.V, **bench.MODEL_CONF).to(dev).train()
par = os.environ.get("PAR", "1") == "1"
eng = TrainEngine(model, lr=1e-3, grad_clip=5.0, static_shapes=True, parallel_decoders=par)
T = 998
batch = dict(features=torch.randn(32, T, 80, device=dev), features_length=torch.full((32,), T, dtype=torch.int32, device=dev),
             targets=torch.randint(2, 3000, (32, 30), dtype=torch.int32, device=dev), targets_length=torch.full((32,), 30, dtype=torch.int32, device=dev))
eng.step(batch); torch.cuda.synchronize()
eng.capture(batch, warmup=1)
for i in range(3):
    torch.cuda.synchronize(); torch.distributed.barrier(); t0 = time.perf_counter()
    eng._graph.replay(); torch.cuda.synchronize(); t1 = time.perf_counter()
    eng.reducer(); torch.cuda.synchronize(); t2 = time.perf_counter()
    eng.optimizer.step(lr_from_device=True); eng.seed_counter.add_(1); torch.cuda.synchronize(); t3 = time.perf_counter()
    if rank == 0:
        print(f"par={par} replay {1e3*(t1-t0):.1f} ms  allreduce {1e3*(t2-t1):.1f} ms  optimizer {1e3*(t3-t2):.1f} ms", flush=True)
torch.distributed.destroy_process_group()
