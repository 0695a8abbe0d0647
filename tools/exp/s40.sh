cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "unet or pool or UNet or cnn" 2>&1 | tail -3
for i in 1 2; do
python bench.py --only unet 2>/dev/null | tail -1 | python -c "
import json,sys
d=list(json.load(sys.stdin).values())[0]
print('unet', d['ms_per_step'], d.get('eager_ms_per_step'))"
done
