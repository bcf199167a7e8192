"""Reads the one record of Lasagne's ``get_all_layers`` order that the reference holds -- the ``print_network`` OUTPUT of
``adenet_v3`` stored in avletters/avletters_training.ipynb (the cell that calls ``adenet_v3.create_model(ae, diff_ae, ...,
250, window, 26, fusiontype)``) -- into tests/golden/adenet_v3_layers.json: layer names in topological order with their
output shapes, plus the call's lstm_size / classes arguments.  Data only (a recorded program output), no source text.

    python tests/golden/make_adenet_v3_layers.py [/root/reference]
"""
import json
import os
import re
import sys


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    nb = json.load(open(os.path.join(ref, "avletters", "avletters_training.ipynb")))
    found = None
    for i, cell in enumerate(nb["cells"]):
        src = "".join(cell.get("source", ""))
        if cell["cell_type"] == "code" and "adenet_v3.create_model" in src and "print_network" in src:
            text = "".join("".join(o.get("text", "")) for o in cell.get("outputs", []) if o.get("output_type") == "stream")
            layers = []
            for line in text.splitlines():
                m = re.match(r"\[L\] (\S+): \((.*)\)$", line.strip())
                if m:
                    shape = [None if t.strip() == "None" else int(t) for t in m.group(2).split(",") if t.strip()]
                    layers.append({"name": m.group(1), "shape": shape})
            # the positional arguments behind the shape / variable pairs: lstm_size, window, output_classes, fusiontype
            args = re.search(r"\n\s*(\d+),\s*window,\s*(\d+),", src)
            found = {"source": "avletters/avletters_training.ipynb cell %d (recorded output of print_network)" % i,
                     "lstm_size_argument": int(args.group(1)), "output_classes_argument": int(args.group(2)), "layers": layers}
    assert found and len(found["layers"]) > 20, "the notebook cell was not found"
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "adenet_v3_layers.json")
    with open(out, "w") as f:
        json.dump(found, f, indent=1)
    print("wrote", out, len(found["layers"]), "layers")


if __name__ == "__main__":
    main()
