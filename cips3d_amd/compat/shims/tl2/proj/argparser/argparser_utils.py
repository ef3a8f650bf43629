"""tl2.proj.argparser.argparser_utils (train.py:593, 598; gen_images.py / eval_fid.py: add_argument_int)"""
import argparse


def _str2bool(v):
    if isinstance(v, bool):
        return v
    if str(v).lower() in ("yes", "true", "t", "y", "1"):
        return True
    if str(v).lower() in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError(f"boolean value expected, got {v!r}")


def add_argument_bool(parser, name, default=False, help=''):
    parser.add_argument(f"--{name}", type=_str2bool, nargs='?', const=True, default=default, help=help)


def add_argument_int(parser, name, default=0, help=''):
    parser.add_argument(f"--{name}", type=int, default=default, help=help)


def add_argument_str(parser, name, default='', help=''):
    parser.add_argument(f"--{name}", type=str, default=default, help=help)


def add_argument_float(parser, name, default=0., help=''):
    parser.add_argument(f"--{name}", type=float, default=default, help=help)


def print_args(args):
    print("args:\n" + "\n".join(f"  {k}: {v}" for k, v in sorted(vars(args).items())))
