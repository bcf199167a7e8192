"""A LEARNABLE synthetic stand-in for the AVLetters tri-modal set (no dataset ships with the reference; §8d's
generator draws unlearnable noise): 26 classes, utterances of 12-40 frames, three 1200-pixel streams each carrying a
class-dependent temporal signal under per-pixel noise, written in the reference's schema-1 files (App. B / C) so that the
package's own runner (``ip_avsr_amd.runners.nstream`` = reference runners/3stream.py:135-427) trains on it unchanged.

    520 training utterances (20 "speakers" x 26 letters), 260 validation (10 speakers), 52 test (2 speakers)
    stream k frame t of an utterance of class c, length L:
        x = a_k * ( P_k[c] * sin(pi * (t + .5) / L) + Q_k[c] * ((t + .5) / L - .5) ) + speaker offset + N(0, 1)
    then per-frame z-normalisation (what ``samplewisenormalize`` does in the real pipeline; done here so that the .ini can
    leave every preprocessing switch off and the three arms read identical float32 inputs)

Everything is a function of ``seed`` (default 1234).  Used by tests/test_gpu_accuracy.py and
profiles/scripts/accuracy_explore.py."""
import os

import numpy as np
import scipy.io as sio

CLASSES, D, ENC = 26, 1200, (2000, 1000, 500, 50)
SPEAKERS = dict(train=range(1, 21), val=range(21, 31), test=range(31, 33))


def build(root, seed=1234, amplitude=(0.16, 0.12, 0.10), D=D, enc=ENC, lstm_size=250, num_epoch=10, epochsize=20, batchsize=26,
          learning_rate=1e-3, fusiontype="concat", windowsize=9, validation_window=6):
    """Writes stream{1,2,3}.mat, ae{1,2,3}.mat, {train,val,test}.txt and 3stream.ini under ``root``; returns the .ini path."""
    rng = np.random.RandomState(seed)
    n_spk = max(max(v) for v in SPEAKERS.values())
    subjects = np.repeat(np.arange(1, n_spk + 1), CLASSES)
    labels = np.tile(np.arange(CLASSES), n_spk)
    n = len(subjects)
    lens = rng.randint(12, 41, size=n)
    lens[0] = 40
    lens[CLASSES * 20] = 40                                    # the validation split pads to 40 too
    total = int(lens.sum())
    starts = np.r_[0, np.cumsum(lens)[:-1]]
    for k in range(3):
        P, Q = rng.normal(size=(CLASSES, D)), rng.normal(size=(CLASSES, D))
        spk = 0.3 * rng.normal(size=(n_spk + 1, D))
        X = rng.normal(size=(total, D)).astype(np.float32)
        for u in range(n):
            L, c = lens[u], labels[u]
            ph = (np.arange(L) + 0.5) / L
            sig = amplitude[k] * (np.sin(np.pi * ph)[:, None] * P[c][None, :] + (ph - 0.5)[:, None] * Q[c][None, :])
            X[starts[u]:starts[u] + L] += (sig + spk[subjects[u]][None, :]).astype(np.float32)
        X = (X - X.mean(1, keepdims=True)) / X.std(1, keepdims=True)
        sio.savemat(os.path.join(root, "stream%d.mat" % (k + 1)),
                    dict(dataMatrix=X.astype(np.float32), targetsVec=np.repeat(labels, lens)[:, None].astype("float64") + 1,
                         videoLengthVec=lens[:, None].astype("float64"), subjectsVec=subjects[:, None].astype("float64")))
        dims = (D,) + tuple(enc)                               # DBN-like encoder: N(0, 0.01) weights, zero biases
        ae = {}
        for i, (a, b) in enumerate(zip(dims[:-1], dims[1:])):
            ae["w%d" % (i + 1)] = rng.normal(0, 0.01, (a, b)).astype(np.float32)
            ae["b%d" % (i + 1)] = np.zeros((1, b), np.float32)
        sio.savemat(os.path.join(root, "ae%d.mat" % (k + 1)), ae)
    for k, ids in SPEAKERS.items():
        open(os.path.join(root, k + ".txt"), "w").write(",".join(str(i) for i in ids))
    stream = """
[stream{k}]
data = {root}/stream{k}.mat
imagesize = 30,40
model = {root}/ae{k}.mat
input_dimensions = {D}
shape = {shape}
nonlinearities = rectify,rectify,rectify,linear
reorderdata = False
diffimage = False
meanremove = False
samplewisenormalize = False
featurewisenormalize = False
"""
    ini = "".join(stream.format(k=k, root=root, D=D, shape=",".join(str(e) for e in enc)) for k in (1, 2, 3)) + """
[lstm_classifier]
fusiontype = {fusion}
weight_init = glorot
use_peepholes = False
windowsize = {win}
output_classes = {C}
output_classnames = {names}
lstm_size = {H}
matlab_target_offset = True

[training]
validation_window = {vw}
num_epoch = {ne}
learning_rate = {lr}
epochsize = {es}
batchsize = {bs}
train_subjects_file = {root}/train.txt
val_subjects_file = {root}/val.txt
test_subjects_file = {root}/test.txt
""".format(fusion=fusiontype, win=windowsize, C=CLASSES, names=",".join("abcdefghijklmnopqrstuvwxyz"), H=lstm_size, vw=validation_window,
           ne=num_epoch, lr=learning_rate, es=epochsize, bs=batchsize, root=root)
    path = os.path.join(root, "3stream.ini")
    open(path, "w").write(ini)
    return path
