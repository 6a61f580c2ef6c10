"""Which ATen / runtime operations a darts_vqa weight step launches beside the HIP library's kernels: one step under
torch.profiler, ops grouped by name and input shape (count, device time).  Usage: python tools/glue_trace.py [batch] [--sink]"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from lct_vqa_amd import ops
from lct_vqa_amd.darts_vqa import config as dconfig, factory
from lct_vqa_amd.basic_vqa import config as bconfig
from lct_vqa_amd.optim import FusedAdam
B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 16
dev = torch.device('cuda', 0)
dconfig.DEVICE = bconfig.DEVICE = dev
a = types.SimpleNamespace(unified=False, embed_size=512, word_embed_size=300, num_layers=1, hidden_size=512, arch_type='darts',
                          learn_rate=1e-3, step_size=10, gamma=0.1, arch_learn_rate=6e-4, arch_wt_decay=1e-3, qst_only=False)
ds = types.SimpleNamespace(qst_vocab=types.SimpleNamespace(vocab_size=17858), ans_vocab=types.SimpleNamespace(vocab_size=3129))
torch.manual_seed(10)
model = factory.get_vqa_model(a, ds).to(dev).train()
opt = FusedAdam(model.parameters(), lr=1e-3, flatten=True)
ops.grad_sink('--sink' in sys.argv)
img = torch.randn(B, 3, 224, 224, device=dev)
qst = torch.randint(17858, (B, 14), device=dev); qst[:, 0] = 2
lab = torch.randint(3129, (B,), device=dev)
def step():
    opt.zero_grad()
    loss = model._loss(img, qst, lab, False)
    loss.backward()
    opt.step(max_grad_norm=5.0)
for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.key.startswith('aten::') and e.device_time_total > 0:
        rows.append((e.count, e.device_time_total / 1e3, e.key, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
print('count  device_ms  op  shapes')
for r in rows[:45]:
    print(f'{r[0]:6d} {r[1]:9.3f}  {r[2]:28s} {r[3]}')
print('\nby call count (every aten op, device time or not)')
cnt = {}
for e in prof.key_averages(group_by_input_shape=True):
    if e.key.startswith('aten::'):
        k = (e.key, str(e.input_shapes)[:90])
        cnt[k] = cnt.get(k, 0) + e.count
for (k, sh), c in sorted(cnt.items(), key=lambda kv: -kv[1])[:40]:
    print(f'{c:6d}  {k:28s} {sh}')
print('\nmemcpy / memset events')
mc = {}
for e in prof.events():
    n = e.name
    if 'emcpy' in n or 'emset' in n or 'copyBuffer' in n:
        mc[n] = mc.get(n, 0) + 1
for n, c in sorted(mc.items(), key=lambda kv: -kv[1])[:10]:
    print(f'{c:6d}  {n}')
