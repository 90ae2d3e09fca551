set -e
cd $GRAFT_REPO_ROOT
for v in "-DWF_NS=6 -DWF_PD=4" "-DWF_NS=6 -DWF_PD=3" ""; do
  echo "=== variant: [$v]"
  SSD_HIPCC_FLAGS="$v" python -m objectdetection_ssd_amd.build --force > /dev/null 2>&1
  timeout -k 5 200 python tools/wf_stamps.py 2>&1 | grep "^conv" | cut -c1-170
  timeout -k 10 300 python tools/wino_bench.py 5 2>&1 | grep "^conv[123]" | cut -c1-150
done
