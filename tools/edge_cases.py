"""Zero-size and odd-argument calls through the facade: each must succeed."""
import torch, numpy as np, sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
import embodied_amd as emb
from embodied_amd.core.driver import mask_actions
from embodied_amd.core import streams
dev='cuda'
FAILED = []
def t(name, fn):
    try:
        r=fn(); torch.cuda.synchronize(); print('ok  ', name, getattr(r,'shape',None))
    except Exception as e:
        FAILED.append(name)
        print('FAIL', name, type(e).__name__, str(e)[:100])
t('obs_stack n=0', lambda: emb.ops.obs_stack(torch.zeros((0,8,8,4),dtype=torch.uint8,device=dev), layout='channels_first', dtype=torch.bfloat16, scale=1/255))
t('obs_stack 1x1x1', lambda: emb.ops.obs_stack(torch.ones((2,1,1,1),dtype=torch.uint8,device=dev), layout='channels_first', dtype=torch.float32, scale=1.0))
t('mask 0 rows', lambda: mask_actions(torch.zeros((0,3),device=dev), torch.zeros(0,dtype=torch.bool,device=dev)))
t('mask row_elems 0', lambda: mask_actions(torch.zeros((4,0),device=dev), torch.zeros(4,dtype=torch.bool,device=dev)))
table=torch.arange(40,dtype=torch.float32,device=dev).reshape(10,4)
t('rows_gather empty', lambda: emb.ops.rows_gather(table, np.zeros(0,np.int32)))
t('rows_gather', lambda: emb.ops.rows_gather(table, np.array([3,3,9],np.int32)))
t('window count 0', lambda: streams.window(torch.zeros((2,5,3),device=dev), 2, 0))
t('window_batch count 0', lambda: streams.window_batch({'a': torch.zeros((2,5,3),device=dev)}, 1, 0)['a'])
rep = emb.Replay(length=2, capacity=4, chunksize=3)
for i in range(6): rep.add({'x': np.float32(i), 'is_first': False, 'is_last': False}, worker=10**9 + (i%2))
t('huge worker ids', lambda: rep.sample(3)['x'])
rep2 = emb.Replay(length=2, capacity=4, chunksize=3)
for i in range(6): rep2.add({'x': np.float32(i), 'is_first': False, 'is_last': False}, worker=-5)
t('negative worker id', lambda: rep2.sample(3)['x'])
t('director T=1', lambda: emb.scans.director_score(torch.zeros((0,4),device=dev), torch.ones((1,4),device=dev), torch.ones((1,4),device=dev)))
t('split_traj', lambda: emb.scans.split_traj(torch.zeros((16,3,2),device=dev), 8))
env = emb.envs.synthetic.SyntheticBatchEnv(1, shape=(4,4,1))
drv = emb.Driver(batch_env=env, device='cuda'); r=emb.Replay(length=3,capacity=10,chunksize=4); drv.on_step(r.add); drv.reset()
t('driver 1 env', lambda: (drv(lambda c,o: (c, {'action': torch.zeros(1,dtype=torch.int32,device=dev)}, {}), steps=12), r.sample(2)['image'])[1])
