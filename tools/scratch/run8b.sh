echo "== env set inside python after import torch, before first cuda call"
python - <<'PY' 2>&1 | grep "eager G_reg vs replay\|replay twice\|Error" | head -4
import os, torch
os.environ['DEBUG_CLR_GRAPH_PACKET_CAPTURE'] = '0'
exec(open('tools/scratch/dbg_loop8.py').read())
PY
echo "== env set after torch.cuda.init()"
python - <<'PY' 2>&1 | grep "eager G_reg vs replay\|replay twice\|Error" | head -4
import os, torch
torch.cuda.init(); torch.zeros(1, device='cuda')
os.environ['DEBUG_CLR_GRAPH_PACKET_CAPTURE'] = '0'
exec(open('tools/scratch/dbg_loop8.py').read())
PY
