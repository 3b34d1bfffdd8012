for t64 in 256 320 448 512 768; do for t32 in 512 768 1024; do
echo -n "T64=$t64 T32=$t32 : "; SAR_WG_T64=$t64 SAR_WG_T32=$t32 python bench.py --no-also --no-cpu-baseline --no-roofline --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done; done
for mc in 4 8 32; do echo -n "MINC=$mc : "; SAR_WG_MINC=$mc SAR_WG_T64=512 SAR_WG_T32=1024 python bench.py --no-also --no-cpu-baseline --no-roofline --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; done
