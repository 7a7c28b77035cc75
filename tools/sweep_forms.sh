#!/bin/bash
# usage (GPU box): tools/sweep_forms.sh "<cfgs>"  -- launch time of the two SpMV entry points against the kernel forms
# (key: one or two waves per token block; value: 4-wave / 8-wave workgroups), both FMA engines
CFGS=${1:-"c2 c3 c4 c5"}
for eng in valu mfma; do for ks in 1 2; do for vs in 1 2; do
  echo -n "engine=$eng key_split=$ks value_split=$vs  "
  MUSTAFAR_FMA_ENGINE=$eng MUSTAFAR_KEY_SPLIT=$ks MUSTAFAR_VALUE_SPLIT=$vs python tools/microbench.py --cfg $CFGS --rows 1 --iters 30 2>&1 | grep cfg | sed 's/.*"cfg": "\([a-z0-9]*\)".*"key_us": \([0-9.]*\), "value_us": \([0-9.]*\).*/\1 k \2 v \3 |/' | tr "\n" " "
  echo
done; done; done
