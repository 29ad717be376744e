"""Benchmark of the VQA training step on MI355X (BASELINE.json: "train samples/sec (node) at 1/2/4/8 MI355X").

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run, one rank per GPU)

A step = the reference's training step (vqa_model.py:248-266) over one batch of synthetic VQA-v2-shaped input that
is already resident in HBM: zero_grad -> batch flip -> forward (CNN + GRU + attention head) -> sum-BCE against soft
scores -> backward -> (bucketed RCCL gradient all-reduce, SUM) -> clip_grad_norm 0.25 -> Adamax.
Workload at N=1 = BASELINE.json configs[1]: bf16 training step, batch 256 per GPU, 224x224 images, 14 tokens,
3129 answers, H=1024, V=10000 (weak scaling: 256 per GPU at every N).

Rank 0 prints ONE JSON line; at N=1 it also carries
  roofline     : the dominant kernel of the step, timed live with HIP events on the launching stream,
  cpu_baseline : the CPU oracle's fwd+bwd step (config 0: fp32, batch 32) timed on this box's host cores.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec); ~6.3 TB/s achievable
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16
MFMA_F32_PEAK_TFLOPS = 157.3
GFLOP_PER_SAMPLE_TRAIN = 5.22   # 3 x (1.548 + 0.191) GFLOP, algorithm as written


def make_fixture(tmp, vocab=10000, classes=3129):
    """Question JSON whose vocabulary has exactly `vocab` rows (2 specials + 'What' + vocab-3 words seen twice) and a
    label2ans pickle with `classes` answers (the reference's constructor reads both: gru.py:76, vqa_model.py:49)."""
    import pickle
    letters = 'abcdefghijklmnopqrstuvwxyz'

    def word(i):
        s = ''
        for _ in range(4):
            s += letters[i % 26]
            i //= 26
        return 'w' + s

    words = [word(i) for i in range(vocab - 3)]
    qs = []
    for rep in range(2):
        for i in range(0, len(words), 13):
            qs.append({'question': 'what ' + ' '.join(words[i:i + 13])})
    qpath, lpath = os.path.join(tmp, 'questions.json'), os.path.join(tmp, 'label2ans.pkl')
    with open(qpath, 'w') as f:
        json.dump({'questions': qs}, f)
    with open(lpath, 'wb') as f:
        pickle.dump([f'answer{i}' for i in range(classes)], f)
    return qpath, lpath


def host_cores():
    """(physical cores of the machine, logical CPUs this process may run on)."""
    phys = set()
    try:
        pid = cid = None
        for ln in open('/proc/cpuinfo'):
            if ln.startswith('physical id'):
                pid = ln.split(':')[1].strip()
            elif ln.startswith('core id'):
                cid = ln.split(':')[1].strip()
            elif not ln.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    return len(phys) or (os.cpu_count() or 1), usable


def cpu_baseline(hidden, classes, vocab, seconds=12.0):
    """The CPU oracle's training step (fp32, batch 32 = BASELINE.json configs[0]) on this box's host cores:
    fwd+bwd on the job's CPU share (16 threads per GPU on the pool's boxes; the machine's core count is reported beside
    it), the same with clip_grad_norm + Adamax (vqa_model.py:259, :266), and one thread for a per-core figure."""
    import torch
    from oracle import vqa_oracle as O
    from vqa_amd import synth
    phys, usable = host_cores()
    threads = min(16, usable)                    # the box's CPU share for one GPU is 16
    B = 32
    sd = synth.state_dict(42, hidden, classes, vocab)
    img, q, t = synth.images(1234, B, 224), synth.questions(1234, B, 14, vocab), synth.targets(1234, B, classes)
    params = {k: v for k, v in sd.items() if O.is_param(k)}

    def timed(fn, warm, most, budget):
        times, t_end = [], None
        for i in range(warm + most):
            t0 = time.perf_counter()
            fn()
            dt = time.perf_counter() - t0
            if i >= warm:
                times.append(dt)
            if i == warm - 1 or (warm == 0 and i == 0):
                t_end = time.perf_counter() + budget
            if t_end is not None and len(times) >= 2 and time.perf_counter() > t_end:
                break
        times.sort()
        return times[len(times) // 2], len(times)

    def fwd_bwd():
        return O.train_step_grads(sd, img, q, t, 2, 'max')

    state = {}

    def full_step():
        _, _, grads, _ = fwd_bwd()
        _, clipped = O.clip_grad_norm(grads)
        O.adamax_step(params, clipped, state)

    torch.set_num_threads(threads)
    med, n = timed(fwd_bwd, 2, 10, seconds)
    med_full, n_full = timed(full_step, 1, 4, seconds / 3)
    torch.set_num_threads(1)
    med_1, n_1 = timed(fwd_bwd, 0, 2, 0.0)
    torch.set_num_threads(threads)
    return {'value': round(B / med, 2), 'unit': 'samples/s', 'cores': threads, 'kind': 'port',
            'sample': f'oracle fwd+bwd, fp32, batch {B}, 224x224, T=14, C={classes}, H={hidden}; median of '
                      f'{n} steps after 2 warm-up',
            'with_clip_adamax': round(B / med_full, 2), 'with_clip_adamax_steps': n_full,
            'one_thread': round(B / med_1, 2), 'one_thread_steps': n_1,
            'host_physical_cores': phys, 'host_usable_cpus': usable}


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes (scripts/pmc_traffic.sh: separate
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this bench; counters are in KB).  gfx950 correction of
    MI355X_MICROARCH.md (HBM): FETCH_SIZE tallies the 128-byte requests of 16-byte-per-lane reads at 64 bytes, so it is
    doubled; WRITE_SIZE is exact for 16/8-byte streaming stores.  None when no profile is committed for the kernel."""
    for name in ('r03_pmc_traffic.json', 'r02_pmc_traffic.json'):
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', name)
        try:
            rec = json.load(open(path))[kernel]
            return round((2.0 * rec['FETCH_SIZE'] + rec['WRITE_SIZE']) * 1024.0 / rec['launches'])
        except (OSError, KeyError, ValueError, ZeroDivisionError):
            continue
    return None


def pmc_mfma(kernel):
    """MFMA-pipe busy fraction of a kernel family from the committed counter passes (scripts/pmc_mfma.sh ->
    profiles/r03_pmc_mfma.json: SQ_VALU_MFMA_BUSY_CYCLES over the SIMD-cycles of the launches), or None."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r03_pmc_mfma.json')
    alias = {'conv3x3_wgrad_deep': 'conv3x3_wgrad_deep(Cin>=32,Cout>=64)', 'conv3x3_wgrad': 'conv3x3_wgrad_shallow'}
    try:
        return json.load(open(path))[alias.get(kernel, kernel)]['mfma_busy_frac']
    except (OSError, KeyError, ValueError):
        return None


# C-ABI entry point -> the device kernel (template family) that does its work; vqa_conv3x3_fwd launches one of two
# kernels (its tag says which): the single-chunk persistent one (Cin <= 32: HBM-bound) or the multi-chunk tile kernel
# (Cin >= 64: MFMA-bound) - two regimes, priced separately
KERNEL_OF = {'vqa_conv3x3_wgrad': 'conv3x3_wgrad', 'vqa_gemm': 'gemm_bf16', 'vqa_bn_bwd': 'bn_bwd',
             'vqa_bn_act_pool_fwd': 'bn_act_pool_fwd', 'vqa_conv3x3_first_fwd': 'conv3x3_first_fwd',
             'vqa_conv3x3_first_wgrad_bn': 'conv3x3_first_wgrad'}
PMC_NAME = {'conv3x3_igemm_p': 'conv3x3_igemm_p', 'conv3x3_igemm_tile': 'conv3x3_igemm_tile'}
# SURVEY.md 8(d) floor: every conv output y, every block output a and their two gradients written once and read once,
# (3,131,328 + 1,982,876) elements per sample at 224x224 / padding 2, 2 bytes each in bf16
ACT_ELEMS_PER_SAMPLE_224 = 3131328 + 1982876


def price(r, peak_tflops):
    """One family's launches against both roofs -> dict with the binding one."""
    sec = r['ms'] * 1e-3
    gbs, tfl = r['bytes'] / sec / 1e9, r['flops'] / sec / 1e12
    bound = 'mfma' if r['flops'] / (peak_tflops * 1e12) > r['bytes'] / (HBM_PEAK_GBS * 1e9) else 'hbm'
    out = {'bound': bound, 'launches_per_step': r['n'], 'avg_launch_ms': round(r['ms'] / r['n'], 4),
           'kernel_ms_per_step': round(r['ms'], 3), 'tflops': round(tfl, 2), 'gbs': round(gbs, 1),
           'algorithmic_bytes_per_launch': round(r['bytes'] / r['n']),
           'algorithmic_gflop_per_launch': round(r['flops'] / r['n'] / 1e9, 3)}
    if bound == 'hbm':
        out.update(achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit='GB/s', frac=round(gbs / HBM_PEAK_GBS, 4))
    else:
        out.update(achieved=round(tfl, 2), peak=peak_tflops, unit='TFLOP/s', frac=round(tfl / peak_tflops, 4))
    return out


ROOFLINE_STEPS = 5


def roofline_leg(trainer, batch, precision='bf16'):
    """ROOFLINE_STEPS extra, instrumented steps (averaged per launch): HIP events around every C-ABI call on the launching stream (question encoder
    and weight gradients kept on the main stream for this step, so no launch is timed under a concurrent kernel).
    `roofline` = the device kernel with the largest total time, priced against the roof that binds the sum of ITS
    launches; `roofline_by_kernel` = the same for every kernel family that carries algorithmic work."""
    from vqa_amd import _lib as L
    from vqa_amd import cnn as CN
    from vqa_amd import vqa_model as VM
    peak = MFMA_BF16_PEAK_TFLOPS if precision == 'bf16' else MFMA_F32_PEAK_TFLOPS
    side, VM.SIDE_STREAM = VM.SIDE_STREAM, False
    hside, VM.HEAD_WGRAD_STREAM = VM.HEAD_WGRAD_STREAM, False
    wside, CN.WGRAD_STREAM = CN.WGRAD_STREAM, False
    try:
        trainer.step(*batch)
        L.profile_start()
        for _ in range(ROOFLINE_STEPS):
            trainer.step(*batch)
        prof = L.profile_stop()
    finally:
        VM.SIDE_STREAM, VM.HEAD_WGRAD_STREAM = side, hside
        CN.WGRAD_STREAM = wside
    # the same launches in the same order every step: launch i of an entry point = the mean of its ROOFLINE_STEPS
    # records (a single step's figure moves by +-3 % from run to run)
    avg = {}
    for name, recs in prof.items():
        c = len(recs) // ROOFLINE_STEPS
        if c * ROOFLINE_STEPS == len(recs):
            avg[name] = [(sum(recs[i + j * c][0] for j in range(ROOFLINE_STEPS)) / ROOFLINE_STEPS, recs[i][1])
                         for i in range(c)]
        else:       # (an entry point not called the same number of times in every step: per-step totals stay right)
            w = 1.0 / ROOFLINE_STEPS
            avg[name] = [(ms * w, None if tag is None else dict(tag, flops=tag['flops'] * w, bytes=tag['bytes'] * w, _w=w))
                         for ms, tag in recs]
    prof = avg
    totals = {k: sum(ms for ms, _ in v) for k, v in prof.items()}
    rows, fam = {}, {}
    for name, recs in prof.items():
        for ms, tag in recs:
            family = (tag or {}).get('kernel') or KERNEL_OF.get(name, name)
            for table, key in ((rows, name if not tag else f"{name}[{tag['kind']}]"), (fam, family)):
                r = table.setdefault(key, dict(ms=0.0, n=0, flops=0.0, bytes=0.0))
                r['ms'] += ms
                r['n'] += (tag or {}).get('_w', 1)
                if tag:
                    r['flops'] += tag['flops']
                    r['bytes'] += tag['bytes']
    priced = {k: price(r, peak) for k, r in fam.items() if r['bytes'] > 0}
    dom = max(priced, key=lambda k: fam[k]['ms'])
    out = dict(kernel=dom, traffic=pmc_traffic(dom), **priced[dom])
    by_kernel = {k: {kk: v[kk] for kk in ('bound', 'frac', 'achieved', 'unit', 'kernel_ms_per_step', 'launches_per_step')}
                 for k, v in sorted(priced.items(), key=lambda kv: -fam[kv[0]]['ms'])[:9]}
    for k, v in by_kernel.items():
        busy = pmc_mfma(k)
        if busy is not None:
            v['pmc_mfma_busy'] = busy           # rocprofv3 counters: fraction of SIMD-cycles with the MFMA pipe busy
    out['per_launch'] = [f"{name}[{tag['kind']}] {tag['shape']:>18s} {ms:7.3f} ms {tag['flops'] / ms / 1e9:8.1f} TF/s "
                         f"{tag['bytes'] / ms / 1e6:8.1f} GB/s"
                         for name, recs in prof.items() for ms, tag in recs if tag]
    breakdown = {k: round(v['ms'], 3) for k, v in sorted(rows.items(), key=lambda kv: -kv[1]["ms"])[:48]}
    return out, by_kernel, breakdown, sum(totals.values())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    # (defaults: 0.5 s of GPU time.  After the synchronize that opens the timed region the GPU starts with an empty
    # queue and waits for the host through the first two or three steps - the host needs 0.9 ms to issue the head's
    # 0.27 ms of kernels - a fixed ~2 ms that weighs 2.5 % on 20 steps and 0.5 % on 100: 4.22 vs 4.12 ms/step)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--batch', type=int, default=256, help='per-GPU batch (weak scaling)')
    ap.add_argument('--precision', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--hidden', type=int, default=1024)
    ap.add_argument('--image', type=int, default=224)
    ap.add_argument('--tokens', type=int, default=14)
    ap.add_argument('--mode', default='train', choices=['train', 'eval'],
                    help="'eval' = BASELINE.json configs[3]: forward + argmax + soft accuracy, eval-mode BatchNorm "
                         "(use --batch 1024); the headline / driver contract is the default 'train'")
    ap.add_argument('--host-inputs', default='off', choices=['off', 'sync', 'overlap'],
                    help="NOT the headline: every step's batch starts in pinned host memory (the reference's loop does "
                         "batch['image'].to(device), vqa_model.py:250-252).  'sync' copies on the compute stream, "
                         "'overlap' copies the next batch on a second stream while the current step runs")
    ap.add_argument('--serial-streams', action='store_true',
                    help='NOT the headline: question encoder and weight gradients on the main stream for the whole run '
                         '(for rocprofv3 --kernel-trace: per-kernel durations without a concurrent kernel)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--force-buckets', action='store_true',
                    help='NOT the headline: with one rank under torch.distributed.run, run the bucketed RCCL '
                         'all-reduces anyway (the N > 1 step on one GPU: hooks, the collective stream, the joins)')
    ap.add_argument('--breakdown', action='store_true', help='also print the per-kernel time table to stderr')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1 or args.force_buckets:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29741')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        dist.init_process_group('nccl', device_id=dev)     # RCCL over xGMI

    from vqa_amd import synth
    from vqa_amd.optim import Trainer
    import vqa_model
    if args.serial_streams:
        from vqa_amd import cnn as CN
        from vqa_amd import vqa_model as VM
        VM.SIDE_STREAM = VM.HEAD_WGRAD_STREAM = CN.WGRAD_STREAM = False

    classes, vocab = 3129, 10000
    with tempfile.TemporaryDirectory() as tmp:
        qpath, lpath = make_fixture(tmp, vocab, classes)
        torch.manual_seed(42)                                  # module-default inits, as the reference (vqa_model.py:334)
        model = vqa_model.VQA({'word_embd_dim': 300, 'question_hidden_dim': args.hidden, 'n_layers': 1,
                               'train_question_path': qpath}, lpath, 256, 2, 0.0, 'max', 'relu',
                              precision=args.precision)
    assert len(model.gru.word_idx_mappings) == vocab, len(model.gru.word_idx_mappings)
    model = model.to(dev).train()
    trainer = Trainer(model, lr=2e-3, max_norm=0.25, force_buckets=args.force_buckets)
    B = args.batch
    seed = 1234 + rank                                          # every rank its own shard of the global batch
    batch = (synth.images(seed, B, args.image).to(dev), synth.questions(seed, B, args.tokens, vocab).to(dev),
             synth.targets(seed, B, classes).to(dev))

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    if args.mode == 'eval':
        from vqa_amd import ops
        model.eval()

        def step(images, questions, target):
            with torch.no_grad():
                return ops.soft_accuracy_sum(model(images, questions), target)   # device scalar, no host sync
    else:
        step = trainer.step
    if args.host_inputs != 'off':
        host = tuple(t.cpu().pin_memory() for t in batch)
        compute = step
        if args.host_inputs == 'sync':
            def step(*_):
                return compute(*(h.to(dev, non_blocking=True) for h in host))
        else:
            copy_stream = torch.cuda.Stream()
            slots = [tuple(torch.empty_like(t) for t in batch) for _ in range(2)]
            ready = [torch.cuda.Event() for _ in range(2)]
            freed = [torch.cuda.Event() for _ in range(2)]
            state = {'i': 0}

            def upload(k):
                with torch.cuda.stream(copy_stream):
                    copy_stream.wait_event(freed[k])          # the step that used this slot has finished with it
                    for d, h in zip(slots[k], host):
                        d.copy_(h, non_blocking=True)
                    ready[k].record(copy_stream)

            for k in range(2):
                freed[k].record()
            upload(0)

            def step(*_):
                k = state['i'] & 1
                upload(k ^ 1)                                  # next batch travels under this step
                torch.cuda.current_stream().wait_event(ready[k])
                out = compute(*slots[k])
                freed[k].record()
                state['i'] += 1
                return out
    from vqa_amd import optim as _optim
    if args.mode == 'train' and _optim.STEP_PRIORITY:
        # the loop runs on the trainer's own high-priority stream (Trainer.stream: no per-step hand-over to the default
        # stream); the batch was made on the default stream
        trainer.stream.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(trainer.stream)
    for _ in range(args.warmup):
        loss = step(*batch)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step(*batch)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt)
    final_loss = float(loss)
    ms_per_step = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed

    if args.mode == 'eval':
        if dist.is_initialized():
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({'metric': 'eval samples/sec (node)', 'value': round(value, 1), 'unit': 'samples/s',
                              'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                              'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True, 'scaling': 'weak',
                              'vs_baseline': None, 'dtype': args.precision, 'data': 'synthetic',
                              'config': {'workload': 'BASELINE.json configs[3]: eval forward + argmax + soft accuracy, '
                                                     f'{args.precision}, batch {B}/GPU, {args.image}x{args.image}, '
                                                     f'{args.tokens} tokens, H={args.hidden}',
                                         'soft_accuracy_sum': round(final_loss, 3)}}), flush=True)
        return
    result = {
        'metric': 'train samples/sec (node)', 'value': round(value, 1), 'unit': 'samples/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': args.precision, 'data': 'synthetic',
        'config': {'workload': ('BASELINE.json configs[1]' if (args.image, args.tokens, args.hidden, B) == (224, 14, 1024, 256)
                                else 'BASELINE.json configs[4]' if (args.image, args.tokens, args.hidden) == (448, 30, 2048)
                                else 'non-headline variant') + ': VQA training step (fwd+bwd+clip+Adamax), '
                               f'{args.precision}, batch {B}/GPU, {args.image}x{args.image} images, {args.tokens} tokens, '
                               f'{classes} answers, H={args.hidden}, V={vocab}, padding 2, max-pool, dropout 0',
                   'global_batch': world * B, 'per_gpu_batch': B, 'parallelism': f'dp{world}',
                   'weights': 'module-default init, seed 42', 'final_loss': round(final_loss, 3),
                   **({'host_inputs': args.host_inputs} if args.host_inputs != 'off' else {}),
                   **({'force_buckets': True} if args.force_buckets else {})},
        'step_tflops_algorithmic': round(value * GFLOP_PER_SAMPLE_TRAIN / 1e3, 2),
    }
    # every rank runs the instrumented step (it all-reduces like any other); rank 0 reports its own timings
    roof, by_kernel, breakdown, dev_ms = roofline_leg(trainer, batch, args.precision)
    if rank == 0:
        result['roofline'] = roof
        result['roofline_by_kernel'] = by_kernel
        result['device_ms_in_kernels'] = round(dev_ms, 3)
        if args.image == 224:
            # the whole step against both roofs: SURVEY.md 8(d)'s write-once / read-once activation bytes (plus the
            # f16 images and parameters + gradients + Adamax state) per step time, and the algorithmic FLOPs
            act = ACT_ELEMS_PER_SAMPLE_224 * 4 * (2 if args.precision == 'bf16' else 4) * B + 150528 * 2 * B + 194e6
            peak = MFMA_BF16_PEAK_TFLOPS if args.precision == 'bf16' else MFMA_F32_PEAK_TFLOPS
            result['step_roofline'] = {
                'algorithmic_bytes_per_step': round(act), 'hbm_frac': round(act / (ms_per_step * 1e-3) / 8e12, 4),
                'algorithmic_tflops': result['step_tflops_algorithmic'],
                'mfma_frac': round(result['step_tflops_algorithmic'] / world / peak, 4)}
        if args.breakdown:
            print(json.dumps(breakdown, indent=1), file=sys.stderr)
            for ln in roof.pop('per_launch'):
                print(ln, file=sys.stderr)
        roof.pop('per_launch', None)
        if not args.no_cpu_baseline and world == 1:
            result['cpu_baseline'] = cpu_baseline(args.hidden, classes, vocab)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
