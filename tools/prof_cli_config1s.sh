#!/bin/bash
# rocprofv3 kernel statistics of the drop-in CLI itself on the config-1 surrogate's .fa.gz (BGZF): where the device time of a CLI run goes, kernel by kernel.
#   gpurun -- 'bash tools/prof_cli_config1s.sh <tag> [reads]'   -> gpurun_out/<tag>/<tag>_cli_config1s_kernel_stats.csv
TAG=${1:-cli}; N=${2:-100000}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python - <<PY
import os, sys
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
from oatk_amd import synth
cfg = dict(synth.CONFIG1S); rs = synth.MixReadSet(**cfg)
seq, off, lens = rs.slice(0, $N)
synth.write_fasta("/tmp/prof_c1s.fa.gz", seq, off, lens, mode=synth.FA_BGZF)
PY
BIN=$(cd $R/tests && python -c "import cli_util; print(cli_util.CLI_DROPIN)")   # the drop-in CLI the tests build
OATK_DROPIN_LOG=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cli_stats -o t -- $BIN -k 1001 -c 30 -t 32 -o /tmp/prof_c1s_out /tmp/prof_c1s.fa.gz > $O/cli.log 2>&1; echo "rc $?"
f=$(find $O/cli_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${TAG}_cli_config1s_kernel_stats.csv && cut -c1-150 $f | head -32
grep "oatk_dropin\] [a-z_]* *[0-9]" $O/cli.log | head -12
