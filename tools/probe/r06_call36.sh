mkdir -p gpurun_out/r06
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for leg in c5 c2; do
rm -rf gpurun_out/r06/eff_$leg gpurun_out/r06/eff_$leg.log
SG_GRAPHS=0 SG_LAUNCH_LOG=gpurun_out/r06/eff_$leg.log timeout 600 rocprofv3 --kernel-trace -d gpurun_out/r06/eff_$leg -o eff -- python tools/run_leg.py $leg 4 > gpurun_out/r06/eff_$leg.out 2>&1
db=$(find gpurun_out/r06/eff_$leg -name "*.db" | head -1)
python - <<PY
import sqlite3
db = sqlite3.connect("$db")
print([r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')").fetchall()][:60])
try:
    print(db.execute("select * from kernels limit 1").description)
except Exception as e:
    print(e)
PY
python tools/launch_eff.py $db gpurun_out/r06/eff_$leg.log 0 90 > gpurun_out/r06/launch_eff_$leg.md 2>&1
rm -rf gpurun_out/r06/eff_$leg
done
tail -3 gpurun_out/r06/eff_c5.out
head -50 gpurun_out/r06/launch_eff_c5.md
