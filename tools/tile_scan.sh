# GPU box: microbenchmark of the skinny x @ W^T / dy @ W GEMMs at every block tile and both kernel families (tools/gemm_bench.py run()).
cd $GRAFT_REPO_ROOT
for dma in 1 2; do for tile in 11 12 21 22; do
echo "== OE_GEMM_DMA=$dma OE_GEMM_TILE=$tile"
OE_GEMM_DMA=$dma OE_GEMM_TILE=$tile timeout -k 10 120 python - <<'PY' 2>/dev/null
import sys, os
sys.path.insert(0, os.getcwd())
from tools.gemm_bench import run
for kind, m, n, k in (("nt", 7936, 256, 256), ("nn", 7936, 256, 256), ("nt", 7936, 256, 512), ("nt", 7936, 256, 768), ("nn", 7936, 256, 1024), ("nt", 7936, 512, 256), ("nt", 7936, 768, 256), ("nn", 7936, 1024, 256)):
    us, tf = run(kind, m, n, k, 3, reps=20)
    print(f"  {kind} {m}x{n}x{k}: {us:6.1f} us {tf:6.1f} TF/s", flush=True)
PY
done; done
