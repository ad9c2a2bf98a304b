#!/bin/bash
# Developer helper (GPU box): deep_probe under every (pixel tile, waves per block) override of deep2_conv_kernel -> gpurun_out/deep_sweep.txt
cd $GRAFT_REPO_ROOT
out=gpurun_out/deep_sweep.txt; : > $out
for npt in 8 4 2; do for nw in 8 4 2; do
  echo "== npt=$npt nw=$nw" >> $out
  MMVAE_DEEP2_NPT=$npt MMVAE_DEEP2_NW=$nw python tools/deep_probe.py 5120 10 2>/dev/null | awk '{print $1, $3}' | grep -v "^layer" >> $out
done; done
python - <<'PY'
import collections
t=collections.OrderedDict(); cur=None
for l in open('gpurun_out/deep_sweep.txt'):
    l=l.split()
    if l[0]=='==': cur=l[1]+' '+l[2]; continue
    t.setdefault(l[0],{})[cur]=float(l[1])
cols=[f"npt={n} nw={w}" for n in (8,4,2) for w in (8,4,2)]
print(f"{'layer':20s}"+"".join(f"{c.replace('npt=','p').replace(' nw=','w'):>8s}" for c in cols))
for k,v in t.items(): print(f"{k:20s}"+"".join(f"{v.get(c,0):8.1f}" for c in cols))
PY
