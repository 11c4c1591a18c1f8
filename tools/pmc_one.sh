cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_one; rm -rf $out; mkdir -p $out
for c in "$@"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-48)
  timeout -k 5 150 rocprofv3 --pmc $c -d $out/$tag -o p --output-format csv -- python bench.py --tower-only --steps 1 --warmup 1 > $out/$tag.log 2>&1 || echo "pass '$c' failed"
done
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in sorted(glob.glob("gpurun_out/pmc_one/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        a = agg[r["Kernel_Name"]][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
for k in sorted(agg):
    if "stem_pool" in k or "halo64s" in k:
        print(k[:80], "  ".join(f"{c}={v/n:.4g}" for c, (n, v) in sorted(agg[k].items())))
PY
