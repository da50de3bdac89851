cd $GRAFT_REPO_ROOT
python - <<'PY'
import torch
from tests import full_size_parity as fp
ev, frame, pl, sp = fp.synthetic_sample(4, 100000)
print(fp.compare(ev, frame, pl, sp, nwin=4))
PY
