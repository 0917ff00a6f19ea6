#!/bin/bash
# round 4: SQ counters of the headline kernel for builds of the library (MCMC_HIP_LIB)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r4abpmc; rm -rf $O; mkdir -p $O
for v in ${VARIANTS:-old k0r0 k1r1}; do
  MCMC_HIP_LIB=$PWD/cobaya_amd/csrc/_exp/lib_$v.so timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU -d $O/p_$v -o p -- python bench.py --no-cpu-baseline --no-variants --steps 20 --warmup 4 --cross-check-seconds 0 > /dev/null 2>> $O/err.log
  python - $O/p_$v $v <<'PY'
import sys,sqlite3,glob,collections
db=sqlite3.connect(glob.glob(sys.argv[1]+"/*.db")[0])
tabs=[r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
q="select counter_name, avg(value), count(*) from counters_collection where kernel_name like '%step_inc_kernel%' group by counter_name" 
try:
    rows=db.execute(q).fetchall()
except Exception as e:
    print("views:", [t for t in tabs if 'count' in t.lower() or 'pmc' in t.lower()]); raise
print(sys.argv[2], {r[0]: float("%.5g"%r[1]) for r in rows}, "n", rows[0][2] if rows else 0)
PY
  rm -rf $O/p_$v
done
