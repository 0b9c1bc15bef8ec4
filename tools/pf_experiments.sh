cd $GRAFT_REPO_ROOT
run() { env "$@" python - <<'PY'
import os, sys, numpy as np
sys.path.insert(0, "spark-tts_amd")
import torch
from sparkmi import config as Cf, weights as W
from sparkmi.llm import SparkLLM
cfg = Cf.spark_0p5b_llm()
llm = SparkLLM(cfg, W.SyntheticLLM(cfg), "cuda:0", max_positions=512)
llm.prefill([np.random.Generator(np.random.PCG64(1)).integers(0, cfg.vocab_size, size=128).tolist()]); llm.decode(40); torch.cuda.synchronize()
s = [round(llm.time_kernel("step", iters=100) * 1e3, 1) for _ in range(3)]
print({k: v for k, v in os.environ.items() if k.startswith("SPARKMI_")}, "graph step", s, flush=True)
PY
}
for rep in 1 2; do
run A=1
run SPARKMI_PF_QKV=6
run SPARKMI_PF_QKV=8
run SPARKMI_PF_QKV=2
run SPARKMI_PREFETCH=5 SPARKMI_TUNE2=4194304
run SPARKMI_PREFETCH=4 SPARKMI_TUNE2=4194304
done
