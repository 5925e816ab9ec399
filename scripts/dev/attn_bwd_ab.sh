# vidor.yaml training step (48 pairs x 512 frames) with the fused attention backward on / off
for f in ${FLAGS:-1 0 1 0}; do
VRDONE_FUSED_ATTN_BWD=$f python - <<PY
import os, sys, time, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "scripts"))
from train_step import synthetic_batch
from vrdone_amd import configs, synth
from vrdone_amd.models.maskvrd import MaskVRD
dev = torch.device("cuda")
vcfg = configs.model_config("vidor")
m = synth.load_synthetic_weights(MaskVRD(vcfg, device=dev)).to(dev).train()
data = synthetic_batch(vcfg, configs.input_channels(vcfg), dev, n_pairs=48, seed=0)
ts = []
for it in range(7):
    m.zero_grad(set_to_none=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss = m(data)["total_loss"]; loss.backward()
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print("fused attention backward $f: vidor 48x512 forward+backward", round(1e3 * sorted(ts[2:])[2], 2), "ms; peak memory", round(torch.cuda.max_memory_allocated() / 2**30, 2), "GiB")
PY
done
