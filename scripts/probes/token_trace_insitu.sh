# per-launch durations and gaps of a whole decode token in situ (eager issue, rocprofv3 --kernel-trace): -> gpurun_out/tt/insitu.txt
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p $REPO/gpurun_out/tt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/tt/kt -o kt -- python $REPO/scripts/whole_token_once.py ${1:-512} 6 > $REPO/gpurun_out/tt/kt.log 2>&1
python - <<'PY'
import csv, glob, os, statistics as st
repo=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
f=glob.glob(repo+"/gpurun_out/tt/kt/**/*kernel_trace.csv", recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "tce::" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the last token: the last 161 launches
tok=rows[-161:]
def short(r):
    n=r["Kernel_Name"]
    k="attn" if "attn" in n else ("gemv_norm" if "1024, true" in n else "gemv")
    return k, int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size",0))
out=[]; prev_end=None; agg={}
for r in tok:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    k=short(r); gap=(s-prev_end) if prev_end else 0
    agg.setdefault(k,[]).append((e-s,gap)); prev_end=e
total=int(tok[-1]["End_Timestamp"])-int(tok[0]["Start_Timestamp"])
with open(repo+"/gpurun_out/tt/insitu.txt","w") as o:
    o.write(f"token span {total/1000:.1f} us over {len(tok)} launches\n")
    for k,v in agg.items():
        d=[x[0] for x in v]; g=[x[1] for x in v]
        o.write(f"{k}: n={len(v)} dur med {st.median(d)/1000:.2f} mean {st.mean(d)/1000:.2f} us | gap before it med {st.median(g)/1000:.2f} mean {st.mean(g)/1000:.2f} us\n")
print(open(repo+"/gpurun_out/tt/insitu.txt").read())
PY
rm -rf $REPO/gpurun_out/tt/kt
