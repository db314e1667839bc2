cd $GRAFT_REPO_ROOT
python scripts/dump_scene.py 1000000 1024 60 E > /dev/null
[ -f scripts/_scene/scene.bin ] || python scripts/dump_scene.py 300000 512 50 scene > /dev/null
for rep in 1 2; do for L in libr2hip_b7ce686.so libr2hip.so; do
  echo "== E $L rep $rep: $(R2_SCENE=scripts/_scene/E.bin timeout 300 scripts/cbench 120 r2_gaussian_amd/$L single,stages 2>&1 | grep -E 'BEST|raster\.' | tr -s ' ' | tr '\n' ';' | sed 's/raster\.//g')"
  echo "== scene $L rep $rep: $(timeout 300 scripts/cbench 300 r2_gaussian_amd/$L single,stages,batch 2>&1 | grep -E 'BEST|BATCH|raster\.pre' | tr -s ' ' | tr '\n' ';' | sed 's/raster\.//g')"
done
echo "== scene mvshare0 rep $rep: $(R2_TF_MV_SHARE=0 timeout 300 scripts/cbench 300 r2_gaussian_amd/libr2hip.so batch 2>&1 | grep -E 'BATCH' | tr -s ' ' | tr '\n' ';')"
done
