"""Where an outer batch of optimize_pose_refiner() spends its time: every engine method / host helper the driver calls is wrapped with a
synchronising timer (so the numbers add up to MORE than the un-instrumented batch: the syncs serialise host and device).
    python tools/exp/driver_sections.py [--all_vertex_tiles] [driver flags ...]"""
import importlib, os, sys, time, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
PKG = 'joint-regressor-refinement_amd'
argsmod = importlib.import_module(PKG + '.args')
flags = ['--batch_size', '4096', '--synthetic_batches', '3', '--inner_iters', '100', '--synthetic', '--smpl_dir', '/nonexistent',
         '--j_regressor_init', '/nonexistent'] + sys.argv[1:]
argsmod._LazyArgs._ns = argsmod.get_args(flags)
eng_mod = importlib.import_module(PKG + '.engine')
opt = importlib.import_module(PKG + '.optimize')
utils = importlib.import_module(PKG + '.utils')
T = collections.defaultdict(list)

def wrap(obj, name, label=None):
    fn = getattr(obj, name)
    def w(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); T[label or name].append(time.perf_counter() - t0)
        return r
    setattr(obj, name, w)

for n in ('refine_run', 'refine_run_j_steps', 'pose_disc_backward_params', 'shape_disc_backward_params', 'j_regressor_grad', 'j_step_apply',
          'find_joints_forward', 'set_j_regressor', 'j_support_info', 'set_pose_disc', 'refine_aux_losses', 'camera_prefit', 'set_loss_history'):
    wrap(eng_mod.RefineEngine, n)
wrap(utils, 'evaluate_sums'); wrap(utils, 'move_pelvis'); wrap(eng_mod, 'adam_step')
# a fixed calibration product timed right after every refine_run: does a slow batch's device state slow IT too?
_A = torch.randn(2048, 2048, device='cuda:0')
CAL = []
_rr = eng_mod.RefineEngine.refine_run
ENQ = []
def _rr_cal(self, *a, **k):
    torch.cuda.synchronize(); te = time.perf_counter()
    r = _rr(self, *a, **k)
    ENQ.append((time.perf_counter() - te) * 1e3)          # host time of the call itself (it returns after its last launch)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5):
        _A @ _A
    torch.cuda.synchronize(); CAL.append((time.perf_counter() - t) * 1e3)
    return r
if os.environ.get('JRR_SECTIONS_CAL') == '1':
    eng_mod.RefineEngine.refine_run = _rr_cal
recs = []
t0 = time.perf_counter()
opt.optimize_pose_refiner(log=recs.append)
torch.cuda.synchronize()
print('whole call %.3f s; seconds_batch per batch: %s' % (time.perf_counter() - t0, [round(r['seconds_batch'], 4) for r in recs]))
nb = len(recs)
tot = 0.0
for k, v in sorted(T.items(), key=lambda kv: -sum(kv[1])):
    last = v[len(v) * (nb - 1) // nb:]          # the calls of the last batch
    print(f'{k:32s} calls/batch {len(v) / nb:4.1f}   last batch: {sum(last) * 1e3:8.3f} ms')
    tot += sum(last)
print(f'sum of the wrapped calls, last batch: {tot * 1e3:.3f} ms of {recs[-1]["seconds_batch"] * 1e3:.3f} ms')
if CAL:
    print('host time inside refine_run (ms):', ' '.join('%.2f' % c for c in ENQ))
    print('calibration product after each refine_run (ms):', ' '.join('%.2f' % c for c in CAL))
# per batch: the sections that moved (ms), to tell a slow batch's cause
for k, v in sorted(T.items(), key=lambda kv: -sum(kv[1])):
    per = len(v) // nb
    if per and sum(v) > 1e-3:
        print(f'{k:32s}', ' '.join('%7.2f' % (sum(v[i * per:(i + 1) * per]) * 1e3) for i in range(nb)))
