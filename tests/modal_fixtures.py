"""Synthetic stand-ins for the schema-2 scripts' inputs (SURVEY.md App. B schema 2, App. C): small .mat / .ini / split files
with the reference's keys.  Shared by the host test of the loaders and the GPU end-to-end test of the drivers."""
import os
import pickle

import numpy as np
import scipy.io as sio

D, DCT, CLASSES = 24, 9, 4             # 4 x 6 "images", 3 DCT coefficients + deltas


def _utterances(rng, n, dim, labels, protos):
    lens = rng.randint(5, 11, size=n)
    rows = [protos[labels[u]][None, :] * (0.5 + np.linspace(0, 1, lens[u])[:, None]) + rng.normal(size=(lens[u], dim)) * 0.3
            for u in range(n)]
    return lens, np.concatenate(rows), np.concatenate([np.full(lens[u], labels[u]) for u in range(n)])


def _ae(rng, root, name, d_in=D, as_pickle=False):
    dims = [d_in, 16, 12, 8, 5]
    ws = [rng.normal(0, 0.3, (a, b)).astype(np.float32) for a, b in zip(dims[:-1], dims[1:])]
    bs = [rng.normal(0, 0.05, (b,)).astype(np.float32) for b in dims[1:]]
    path = os.path.join(root, name)
    if as_pickle:
        with open(path, "wb") as f:
            pickle.dump((ws, bs), f)
    else:
        sio.savemat(path, dict([("w%d" % (i + 1), ws[i]) for i in range(4)] + [("b%d" % (i + 1), bs[i][None, :]) for i in range(4)]))
    return path


def make_cuave(root, seed=0):
    """Pre-split CUAVE files (cuave/bimodal_with_val.py:211-222): targets are stored 0-based minus one (the script adds 1)."""
    rng = np.random.RandomState(seed)
    protos, dprotos = rng.normal(size=(CLASSES, D)) * 2, rng.normal(size=(CLASSES, DCT)) * 2
    data, dct = {}, {}
    for pre, n in (("tr", 32), ("val", 12), ("test", 12)):
        labels = np.arange(n) % CLASSES
        lens, X, tv = _utterances(rng, n, D, labels, protos)
        feats = np.concatenate([dprotos[labels[u]][None, :] + rng.normal(size=(lens[u], DCT)) * 0.3 for u in range(n)])
        data[pre + "Data"] = X
        data[pre + "VideoLengthVec"] = lens[:, None].astype("float64")
        data[pre + "TargetsVec"] = (tv - 1)[:, None].astype("float64")          # per FRAME, 0-based minus one (the script adds 1)
        dct[pre + "DctFeatures"] = feats
    sio.savemat(os.path.join(root, "cuave.mat"), data)
    sio.savemat(os.path.join(root, "cuave_dct.mat"), dct)
    ae = _ae(rng, root, "ae.mat")
    ini = """
[data]
images = {root}/cuave.mat
dct = {root}/cuave_dct.mat
imagesize = 4,6

[models]
pretrained = {ae}
fusiontype = adasum
input_dimension = {D}
no_coeff = 3
output_classes = {C}
lstm_size = 10
nonlinearity = rectify

[training]
validation_window = 4
num_epoch = 5
weight_init = glorot
learning_rate = 0.01
epochsize = 4
batchsize = 8
use_peepholes = True
use_blstm = True
use_finetuning = False
""".format(root=root, ae=ae, D=D, C=CLASSES)
    path = os.path.join(root, "cuave.ini")
    open(path, "w").write(ini)
    return path


def _frames_file(root, rng, name, with_iter, one_based=False):
    subjects = np.repeat(np.arange(1, 9), 6)                  # 8 subjects x 6 utterances
    n = len(subjects)
    labels = np.arange(n) % CLASSES
    protos, dprotos = rng.normal(size=(CLASSES, D)) * 2, rng.normal(size=(CLASSES, DCT)) * 2
    lens, X, tv = _utterances(rng, n, D, labels, protos)
    feats = np.concatenate([dprotos[labels[u]][None, :] + rng.normal(size=(lens[u], DCT)) * 0.3 for u in range(n)])
    d = dict(dataMatrix=X, targetsVec=(tv + (1 if one_based else 0))[:, None].astype("float64"), videoLengthVec=lens[:, None].astype("float64"),
             subjectsVec=subjects[:, None].astype("float64"))
    if with_iter:
        d["iterVec"] = (np.arange(n) % 3 + 1)[:, None].astype("float64")        # repetitions 1, 2 -> train; 3 -> test
    sio.savemat(os.path.join(root, name + ".mat"), d)
    sio.savemat(os.path.join(root, name + "_dct.mat"), dict(dctFeatures=feats))
    return X, lens


def make_oulu(root, seed=1):
    rng = np.random.RandomState(seed)
    _frames_file(root, rng, "oulu", with_iter=False)
    for k, ids in (("train", "1,2,3,4,5"), ("val", "6,7"), ("test", "8")):
        open(os.path.join(root, k + ".txt"), "w").write(ids)
    ae, ae_diff = _ae(rng, root, "ae_raw.pkl", as_pickle=True), _ae(rng, root, "ae_diff.mat")
    ini = """
[data]
images = {root}/oulu.mat
dct = {root}/oulu_dct.mat

[models]
pretrained = unused
finetuned = {ae}
finetuned_diff = {ae_diff}
fusiontype = sum
input_dimension = {D}
output_classes = {C}
lstm_size = 6

[training]
learning_rate = 1.0
decay_rate = 0.5
decay_start = 2
do_finetune = False
save_finetune = False
load_finetune = True
load_finetune_diff = True
savemodel = False
num_epoch = 4
epochsize = 4
batchsize = 8
validation_window = 4
train_subjects_file = {root}/train.txt
val_subjects_file = {root}/val.txt
test_subjects_file = {root}/test.txt
""".format(root=root, ae=ae, ae_diff=ae_diff, D=D, C=CLASSES)
    path = os.path.join(root, "oulu.ini")
    open(path, "w").write(ini)
    return path


def make_avletters(root, seed=2, update_rule="sgdm"):
    rng = np.random.RandomState(seed)
    X, lens = _frames_file(root, rng, "avl", with_iter=True)
    # avletters/bimodal.py subtracts 1 from the (MATLAB, 1-based) targets (:351); avletters/trimodal.py uses them as stored
    _frames_file(root, np.random.RandomState(seed), "avl1", with_iter=True, one_based=True)
    diff = np.concatenate([np.vstack([np.zeros((1, D)), np.diff(X[s:s + l], axis=0)])
                           for s, l in zip(np.cumsum(np.r_[0, lens[:-1]]), lens)])
    sio.savemat(os.path.join(root, "avl_diff.mat"), dict(dataMatrix=diff))
    ae, ae_diff = _ae(rng, root, "avl_ae.mat"), _ae(rng, root, "avl_ae_diff.mat")
    common = """
[data]
images = {root}/avl.mat
dct = {root}/avl_dct.mat
diff = {root}/avl_diff.mat

[models]
pretrained = {ae}
finetuned = {ae}
finetuned_diff = {ae_diff}
fusiontype = concat
input_dimension = {D}
no_coeff = 3
output_classes = {C}
lstm_size = 6
""".format(root=root, ae=ae, ae_diff=ae_diff, D=D, C=CLASSES)
    tri = common + """
[training]
learning_rate = 1.0
decay_rate = 0.9
decay_start = 3
do_finetune = False
save_finetune = False
load_finetune = True
load_finetune_diff = True
num_epoch = 4
epochsize = 4
batchsize = 8
"""
    bi = common.replace("avl.mat", "avl1.mat").replace("avl_dct.mat", "avl1_dct.mat") + """
[training]
update_rule = {rule}
learning_rate = 0.05
decay_rate = 0.8
decay_start = 3
t1 = 1
momentum = 0.5
momentum_schedule = 0.7,0.9
validation_window = 4
num_epoch = 5
weight_init = ortho
use_peepholes = False
use_blstm = True
use_finetuning = False
epochsize = 4
batchsize = 8
""".format(rule=update_rule)
    p_tri, p_bi = os.path.join(root, "avl_tri.ini"), os.path.join(root, "avl_bi.ini")
    open(p_tri, "w").write(tri)
    open(p_bi, "w").write(bi)
    return p_tri, p_bi


# ----------------------------------------------------------------------------------------------- round 3: the rest of the family
def make_cuave_subject(root, seed=3):
    """cuave/unimodal_with_val.py: ONE whole-set file with subjectsVec (targets stored minus one: the script adds 1)."""
    rng = np.random.RandomState(seed)
    subjects = np.repeat(np.arange(1, 9), 6)
    n = len(subjects)
    labels = np.arange(n) % CLASSES
    protos = rng.normal(size=(CLASSES, D)) * 2
    lens, X, tv = _utterances(rng, n, D, labels, protos)
    sio.savemat(os.path.join(root, "cuave_all.mat"),
                dict(dataMatrix=X, targetsVec=(tv - 1)[:, None].astype("float64"), videoLengthVec=lens[:, None].astype("float64"),
                     subjectsVec=subjects[:, None].astype("float64")))
    for k, ids in (("train", "1,2,3,4,5"), ("val", "6,7"), ("test", "8")):
        open(os.path.join(root, k + ".txt"), "w").write(ids)
    ae = _ae(rng, root, "ae_uni.mat")
    ini = """
[data]
images = {root}/cuave_all.mat
imagesize = 4,6

[models]
pretrained = {ae}
input_dimension = {D}
output_classes = {C}
lstm_size = 8
nonlinearity = rectify

[training]
validation_window = 4
no_epochs = 4
weight_init = glorot
learning_rate = 0.01
epochsize = 4
batchsize = 8
use_peepholes = False
train_subjects_file = {root}/train.txt
val_subjects_file = {root}/val.txt
test_subjects_file = {root}/test.txt
""".format(root=root, ae=ae, D=D, C=CLASSES)
    path = os.path.join(root, "cuave_uni.ini")
    open(path, "w").write(ini)
    return path


def make_cuave_family(root, seed=0):
    """.ini files of cuave/unimodal_dct_with_val.py, cuave/trimodal_with_val.py and cuave/audio_visual_runner.py on the
    pre-split files of make_cuave (+ an audio file with the same utterance structure)."""
    make_cuave(root, seed)
    rng = np.random.RandomState(seed + 100)
    data = sio.loadmat(os.path.join(root, "cuave.mat"))
    AD = 14
    audio = {}
    for pre in ("tr", "val", "test"):
        n = data[pre + "Data"].shape[0]
        audio[pre + "Data"] = rng.normal(size=(n, AD)) + 0.5 * data[pre + "TargetsVec"]
        audio[pre + "VideoLengthVec"] = data[pre + "VideoLengthVec"]
    sio.savemat(os.path.join(root, "cuave_audio.mat"), audio)
    ae, ae_diff, ae_audio = _ae(rng, root, "f_ae.mat"), _ae(rng, root, "f_ae_diff.pkl", as_pickle=True), _ae(rng, root, "f_ae_audio.mat", d_in=AD)
    data_sec = "[data]\nimages = {root}/cuave.mat\ndct = {root}/cuave_dct.mat\naudio = {root}/cuave_audio.mat\nimagesize = 4,6\n".format(root=root)
    dct = data_sec + """
[models]
no_coeff = 3
output_classes = {C}
lstm_size = 8

[training]
validation_window = 4
no_epochs = 4
weight_init = glorot
learning_rate = 0.01
epochsize = 4
batchsize = 8
use_peepholes = False
use_blstm = True
use_finetuning = False
""".format(C=CLASSES)
    tri = data_sec + """
[models]
finetuned = {ae}
finetuned_diff = {ae_diff}
fusiontype = adasum
input_dimension = {D}
output_classes = {C}
lstm_size = 6

[training]
learning_rate = 1.0
decay_rate = 0.5
decay_start = 2
load_finetune = True
load_finetune_diff = True
num_epoch = 3
epochsize = 3
batchsize = 8
""".format(ae=ae, ae_diff=ae_diff, D=D, C=CLASSES)
    av = data_sec + """
[models]
pretrained = {ae}
pretrained_diff = {ae_audio}
fusiontype = concat
lstm_size = 8
output_classes = {C}
nonlinearity = rectify
input_dimension = {D}
input_dimension2 = {AD}

[training]
validation_window = 4
num_epoch = 4
weight_init = ortho
learning_rate = 0.01
use_peepholes = True
use_blstm = True
use_finetuning = False
epochsize = 4
batchsize = 8
""".format(ae=ae, ae_audio=ae_audio, D=D, AD=AD, C=CLASSES)
    paths = []
    for name, text in (("cuave_dct.ini", dct), ("cuave_tri.ini", tri), ("cuave_av.ini", av)):
        paths.append(os.path.join(root, name))
        open(paths[-1], "w").write(text)
    return paths


def make_oulu_family(root, seed=1):
    """.ini files of oulu/unimodal_with_val.py and oulu/bimodal_with_val.py on the files of make_oulu."""
    make_oulu(root, seed)
    rng = np.random.RandomState(seed + 100)
    ae = _ae(rng, root, "o_ae.mat")
    splits = "train_subjects_file = {root}/train.txt\nval_subjects_file = {root}/val.txt\ntest_subjects_file = {root}/test.txt\n".format(root=root)
    uni = """
[data]
images = {root}/oulu.mat

[models]
pretrained = {ae}
lstm_units = 8
output_classes = {C}
weight_init = glorot
delta_window = 3
nonlinearity = rectify
input_dimension = {D}

[training]
learning_rate = 0.01
no_epochs = 4
use_peepholes = True
epochsize = 4
batchsize = 8
validation_window = 4
""".format(root=root, ae=ae, C=CLASSES, D=D) + splits
    bi = """
[data]
images = {root}/oulu.mat
dct = {root}/oulu_dct.mat

[models]
pretrained = {ae}
fusiontype = sum
output_classes = {C}
lstm_size = 8
delta_window = 3
input_dimensions = {D}
use_peepholes = False
no_coeffs = {DCT}

[training]
learning_rate = 0.01
validation_window = 4
batchsize = 8
epochsize = 4
no_epochs = 3
weight_init = glorot
honour_ini_sizes = True
""".format(root=root, ae=ae, C=CLASSES, D=D, DCT=DCT) + splits
    p_uni, p_bi = os.path.join(root, "oulu_uni.ini"), os.path.join(root, "oulu_bi.ini")
    open(p_uni, "w").write(uni)
    open(p_bi, "w").write(bi)
    return p_uni, p_bi


def make_avletters_family(root, seed=2, update_rule="sgdnm"):
    """.ini files of avletters/bimodal_diff_image.py (schema 2) and avletters/unimodal.py (schema 1, with and without an
    encoder) on the files of make_avletters."""
    make_avletters(root, seed)
    rng = np.random.RandomState(seed + 100)
    ae, ae_diff = _ae(rng, root, "d_ae.mat"), _ae(rng, root, "d_ae_diff.pkl", as_pickle=True)
    diff = """
[data]
images = {root}/avl.mat
diff = {root}/avl_diff.mat

[models]
pretrained = unused
finetuned = {ae}
finetuned_diff = {ae_diff}
fusiontype = adasum
model = adenet_v2_1
input_dimension = {D}
output_classes = {C}
lstm_size = 6

[training]
do_finetune = False
save_finetune = False
load_finetune = True
load_finetune_diff = True
update_rule = {rule}
learning_rate = 0.05
decay_rate = 0.8
decay_start = 2
t1 = 1
momentum = 0.5
momentum_schedule = 0.7,0.9
validation_window = 4
num_epoch = 4
weight_init = ortho
use_peepholes = False
epochsize = 4
batchsize = 8
""".format(root=root, ae=ae, ae_diff=ae_diff, D=D, C=CLASSES, rule=update_rule)
    uni = """
[stream1]
data = {root}/avl1.mat
has_encoder = {enc}
input_dimensions = {D}
imagesize = 4,6
model = {ae}
shape = 16,12,8,5
nonlinearities = rectify,rectify,rectify,linear
reorderdata = True
diffimage = False
meanremove = True
samplewisenormalize = True
featurewisenormalize = {fw}

[lstm_classifier]
output_classes = {C}
output_classnames = a,b,c,d
lstm_size = 8
matlab_target_offset = True
weight_init = glorot
use_peepholes = False
windowsize = 3

[training]
validation_window = 4
num_epoch = 4
learning_rate = 0.01
epochsize = 4
batchsize = 8
"""
    paths = [os.path.join(root, "avl_diff.ini"), os.path.join(root, "avl_uni_enc.ini"), os.path.join(root, "avl_uni_raw.ini")]
    open(paths[0], "w").write(diff)
    open(paths[1], "w").write(uni.format(root=root, ae=ae, D=D, C=CLASSES, enc="True", fw="False"))
    open(paths[2], "w").write(uni.format(root=root, ae=ae, D=D, C=CLASSES, enc="False", fw="True"))
    return paths
