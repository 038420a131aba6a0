"""Golden vectors from the REFERENCE's own config parser (casapose/utils/config_parser.py imports only argparse, configparser and numpy,
so it runs in this container): parse_config() on the reference's config_8.ini / config_13.ini under several command lines.
Run here:  python tests/golden/make_config_golden.py  ->  tests/golden/config_parser_ref.json   (data only; nothing of the reference is stored)"""
import importlib.util
import json
import os
import sys

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

CASES = [
    ["-c", "config/config_8.ini"],
    ["-c", "config/config_13.ini"],
    ["-c", "config/config_8.ini", "--object", "obj_000001", "--batchsize", "1", "--estimate_confidence", "no", "--lr", "0.002"],
    ["-c", "config/config_13.ini", "--use_bpnp_reprojection_loss", "0", "--modelname", "pvnet", "--imagesize", "320", "480"],
    ["--data", "/tmp/x", "--train_vectors_with_ground_truth", "t"],
    [],
]


def jsonable(v):
    import numpy as np
    if isinstance(v, np.ndarray):
        return {"__ndarray__": v.tolist(), "dtype": str(v.dtype)}
    if isinstance(v, (np.integer,)):
        return int(v)
    if isinstance(v, (np.floating,)):
        return float(v)
    if isinstance(v, (list, tuple)):
        return {"__%s__" % type(v).__name__: [jsonable(e) for e in v]}
    return v


def ini_items(argv):
    """the [defaults] section of the case's config file as plain key / value strings (the test writes them into a temporary .ini)"""
    import configparser
    if "-c" not in argv:
        return None
    cp = configparser.ConfigParser()
    cp.read([argv[argv.index("-c") + 1]])
    return {sec: dict(cp.items(sec)) for sec in cp.sections()}


def main():
    spec = importlib.util.spec_from_file_location("ref_config_parser", os.path.join(REF, "casapose/utils/config_parser.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = []
    cwd = os.getcwd()
    os.chdir(REF)   # the config paths of the cases are relative to the reference's root, as its README runs them
    try:
        for argv in CASES:
            sys.argv = ["test_casapose.py"] + argv
            opt = mod.parse_config()
            out.append({"argv": argv, "ini": ini_items(argv), "opt": {k: jsonable(v) for k, v in sorted(vars(opt).items())}})
    finally:
        os.chdir(cwd)
    with open(os.path.join(HERE, "config_parser_ref.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote %d cases, %d options each" % (len(out), len(out[0]["opt"])))


if __name__ == "__main__":
    main()
