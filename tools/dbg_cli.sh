#!/bin/bash
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys; sys.path.insert(0,'tests')
import adversarial as A, ref_lib as R
reads=A.hifi_like(260, 50000, 5000, seed=301 + 77, err=0.001)
R.write_fasta(reads,'/tmp/r.fa')
PY
export OATK_DROPIN_LOG=1
if which gdb >/dev/null; then gdb -batch -ex run -ex bt --args oracle/_ref/syncasm_dropin -k 301 -s 21 -c 6 -t 4 -o /tmp/dev /tmp/r.fa 2>&1 | tail -40; else oracle/_ref/syncasm_dropin -k 301 -s 21 -c 6 -t 4 -o /tmp/dev /tmp/r.fa 2>&1 | tail -5; fi
