cd /root/repo
for m in "" far s1; do echo "== mode '$m'"; ISAC_EIG_TRIDIAG_DIST=$m ISAC_DEBUG=1 python tools/_tridiag_ab.py 2>&1 | grep -v amdgpu.ids | grep "distributed\|n= " | tail -8; done
