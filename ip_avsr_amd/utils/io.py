"""File I/O used by the runners (reference utils/io.py:11-48): subject-split files, ``.mat``
containers and pickled parameter lists.

Checkpoint format: ``save_model_params`` pickles the plain list returned
by ``network.get_all_param_values()`` -- float32 ndarrays in Lasagne ``get_all_params`` order
(layers in topological order; per-gate LSTM matrices) -- so a list written by the reference's
``utils/io.py:40-42`` loads into the same architecture here and vice versa.  Optimiser state is
not part of the file, exactly as in the reference."""
import pickle

import scipy.io as sio


def read_data_split_file(path, sep=","):
    """First line of ``path`` -> list of int subject ids."""
    with open(path) as f:
        return [int(tok) for tok in f.readline().split(sep)]


def load_mat_file(path):
    return sio.loadmat(path)


def save_mat(dict, path):
    print("save matlab file...")
    sio.savemat(path, dict)


def save_model(model, path):
    with open(path, "wb") as f:
        pickle.dump(model, f, protocol=2)


def load_model(path):
    with open(path, "rb") as f:
        try:
            return pickle.load(f)
        except UnicodeDecodeError:           # a Python-2 pickle of ndarrays
            f.seek(0)
            return pickle.load(f, encoding="latin1")


def save_model_params(network, path):
    save_model(network.get_all_param_values(), path)


def load_model_params(network, path):
    network.set_all_param_values(load_model(path))
    return network
