"""how long trainer.EncodedSplit.host_batch takes on THIS host (the prefetch worker's per-batch cost), alone and beside a busy main thread;
prints the CPU environment it ran in.  python tools/host_batch_time.py"""
import json, os, sys, threading, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import nbest_amd  # noqa: F401
from nbest_amd import config as ncfg, inputs, trainer

print("cpu_count %s, affinity %d, torch threads %d, interop %d" % (os.cpu_count(), len(os.sched_getaffinity(0)), torch.get_num_threads(),
                                                                   torch.get_num_interop_threads()))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu.stat"):
    if os.path.exists(f):
        print(f, open(f).read().strip().replace("\n", " | "))
labels = ncfg.LabelSpace.from_json(os.path.join(ROOT, "tests", "golden", "label_space.json"))
vocab = json.load(open(os.path.join(ROOT, "tests", "golden", "text_vocab.json")))
z = np.load(os.path.join(ROOT, "tests", "golden", "case_text.npz"))
memory = dict(label2idx=json.loads(str(z["label2idx"])), idx2label=labels.idx2label)
data = trainer.read_wcn_data(os.path.join(ROOT, "tests", "golden", "valid_512.txt"))
data = tuple(list(x) * 8 for x in data)
opt = types.SimpleNamespace(batchSize=256, tokenizer=inputs.WordPieceTokenizer(vocab), pre_trained_model="bert", tod_pre_trained_model=None,
                            without_system_act=False, add_l2_loss=False, add_segment_ids=True, n_best=5, max_seq_len=None, random_seed=999)
split = trainer.EncodedSplit(data, opt, memory)
lists = trainer.batch_indices(len(split), 256, shuffle=True, seed=5)
stage = trainer.PinnedStage(pin=False)


def timed(tag):
    for rep in range(3):
        t0 = time.time()
        for ix in lists:
            split.host_batch(ix, stage=stage)
        dt = (time.time() - t0) * 1e3 / len(lists)
    print("%s: host_batch %.2f ms per batch" % (tag, dt), flush=True)


timed("default threads")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for ix in lists:
    split.host_batch(ix, stage=stage)
pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(6)
torch.set_num_threads(1)
timed("torch.set_num_threads(1)")
stop = [False]
def spin():
    while not stop[0]:
        time.sleep(0.0002)
th = threading.Thread(target=spin); th.start()
timed("beside a thread polling with sleep(0.2 ms)")
stop[0] = True; th.join()
if os.path.exists("/sys/fs/cgroup/cpu.stat"):
    print(open("/sys/fs/cgroup/cpu.stat").read().strip().replace("\n", " | "))
