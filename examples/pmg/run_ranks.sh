#!/bin/bash
# Launch a driver (pmg_main by default) as one process per GPU and brick (the reference: mpirun -n 8
# with ROCR_VISIBLE_DEVICES=$SLURM_LOCALID, examples/pmg/submit.sh:29, select_gpu.sh:2).
#   usage: run_ranks.sh px,py,pz [driver options ...]
#          PMG_MAIN=.../bin/cg_main run_ranks.sh 2,2,2 --n 32        (cg_main, vector_update_main)
# Rank r runs with RANK=r LOCAL_RANK=r; rank 0 publishes the RCCL communicator id in a temporary file.
set -euo pipefail
dims=$1; shift
IFS=, read -r px py pz <<< "$dims"
n=$((px * py * pz))
here=$(cd "$(dirname "$0")" && pwd)
exe=${PMG_MAIN:-$here/../../pmg-dolfinx_amd/bin/pmg_main}
idfile=$(mktemp -u /tmp/pmg_amd_id.XXXXXX)
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
pids=()
for ((r = n - 1; r >= 0; r--)); do
  if ((r == 0)); then
    RANK=0 LOCAL_RANK=0 "$exe" --ranks "$dims" --id-file "$idfile" "$@" &
  else
    RANK=$r LOCAL_RANK=$r "$exe" --ranks "$dims" --id-file "$idfile" "$@" > /dev/null &
  fi
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait "$p" || rc=$?; done
rm -f "$idfile"
exit $rc
