cd $GRAFT_REPO_ROOT
for shp in "3 30 64 64" "100 100 300 300"; do
python tools/dbg_cell.py /tmp/a.npz $shp
GET_AMD_LIB=$GRAFT_REPO_ROOT/get_amd/lib/libget_hip_e32.so python tools/dbg_cell.py /tmp/b.npz $shp
python - <<P
import numpy as np
a=np.load('/tmp/a.npz'); b=np.load('/tmp/b.npz')
for k in a.files:
    d=np.abs(a[k]-b[k]); bad=np.argwhere(~(d<=1e-5))
    print("$shp", k, 'max', np.nanmax(d), 'nbad', len(bad), 'first', bad[:6].tolist(), 'nan', int(np.isnan(b[k]).sum()))
P
done
