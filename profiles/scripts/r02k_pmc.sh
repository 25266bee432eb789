cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python profiles/scripts/r02_pmc.py /tmp/r02k_pmc 1.0 1000000 --no-columns > gpurun_out/r02k_pmc.txt 2>&1
cp profiles/pmc_traffic.json gpurun_out/r02k_pmc_traffic.json
python - <<'PY'
import json
d=json.load(open("profiles/pmc_traffic.json"))
for k,v in d["detail"].items():
    if k.startswith("k_lift"): print(k, {a:(round(b/1e6,2) if isinstance(b,float) else b) for a,b in v.items()})
PY
