"""The two reporting helpers the runners call (reference utils/plotting_utils.py:226-245 plot_confusion_matrix,
:278-287 print_network, :132-160 plot_validation_cost).  Everything else in that file is visualisation for
notebooks and is out of scope."""
from tabulate import tabulate


def plot_confusion_matrix(conf_mat, headers, fmt='pipe', savefilename=None):
    """Confusion matrix (rows = target, columns = prediction) as a table string; 'pipe' = markdown."""
    rows = [[h] + [int(v) for v in conf_mat[i]] for i, h in enumerate(headers)]
    table = tabulate(rows, headers, tablefmt=fmt)
    if savefilename:
        with open(savefilename, mode='a') as f:
            f.write(table + '\n')
    return table


def print_network(network):
    """One '[L] name: shape' line per parameterised tensor (the reference prints Lasagne layer output shapes;
    here the graph is fixed-function, so the informative part is the parameter table)."""
    for p in network.params:
        print('[L] {}: {}'.format(p.name, p.shape))
    print('[L] total parameters: {}'.format(network.count_params()))


def plot_validation_cost(train_error, val_error, class_rate=None, savefilename=None):
    """Train / validation cost curves (matplotlib, Agg backend; silently skipped when unavailable)."""
    try:
        import matplotlib
        matplotlib.use('Agg')
        import matplotlib.pyplot as plt
    except Exception:
        return None
    epochs = range(1, len(train_error) + 1)
    fig, ax1 = plt.subplots()
    ax1.plot(epochs, train_error, label='train loss')
    ax1.plot(epochs, val_error, label='validation loss')
    ax1.set_xlabel('epoch'); ax1.set_ylabel('loss'); ax1.legend(loc='upper left')
    if class_rate is not None:
        ax2 = ax1.twinx()
        ax2.plot(epochs, class_rate, 'g', label='classification rate')
        ax2.set_ylabel('classification rate')
    if savefilename:
        fig.savefig(savefilename)
    plt.close(fig)
    return savefilename
