python -m pytest tests -x -q -m gpu -k "underfilled or attention_tiny or vitgan or Generator" 2>&1 | tail -3
python - <<'PY'
import torch, os, sys
sys.path.insert(0, os.getcwd())
from feed_forward_vqgan_clip_amd import kernels as K
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
for (M,N,Kd) in [(512,1024,4096),(512,1024,1024),(512,4096,1024),(512,1024,3064),(616,768,9216),(616,3072,2304),(256,1024,4096)]:
    x=torch.randn(M,Kd,device='cuda').half(); w=torch.randn(N,Kd,device='cuda').half(); y=torch.empty(M,N,device='cuda',dtype=torch.half)
    us=t(lambda: K.gemm(x,w,y,M,N,Kd,ldx=Kd,ldw=Kd))
    print(f"NT {M}x{N}x{Kd}: {us:.1f} us  {2*M*N*Kd/us/1e6:.0f} TFLOP/s")
PY
C="--steps 6 --warmup 2 --no-cpu-baseline --no-alt-dtype"
python bench.py $C --model-type vitgan --batch 32 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg3', d['ms_per_step'])"
