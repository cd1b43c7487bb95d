for v in base bk16; do
  if [ "$v" != base ]; then export RECBOX_HIP_LIB=/root/repo/recbox_amd/lib/librecbox_hip_$v.so; else unset RECBOX_HIP_LIB; fi
  echo "== $v"; timeout 300 python profiles/ubench/rocblas_compare.py 2>&1 | tail -7
done
