cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "resnet or ResNet or dropout" 2>&1 | tail -4
python bench.py --only resnet 2>/dev/null | tail -1 | python -c "
import json,sys
d=list(json.load(sys.stdin).values())[0]
print('resnet', d['ms_per_step'], d.get('mode'), d.get('eager_ms_per_step'), d.get('graph_error'))"
