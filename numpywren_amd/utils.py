"""Small helpers with the same contracts as the reference's numpywren/utils.py."""


def convert_to_slice(l):
    """None | int | [stop] | [start, stop] | [start, stop, step] -> slice, in block units
    (same contract as reference numpywren/utils.py:7-29, used by BigMatrix.submatrix)."""
    if l is None:
        return slice(None, None, None)
    if isinstance(l, int):
        return slice(l, l + 1, 1)
    if isinstance(l, slice):
        raise ValueError("Could not convert to slice.")
    n = len(l)
    if n == 1:
        return slice(None, l[0], None)
    if n == 2:
        return slice(l[0], l[1], None)
    if n == 3:
        return slice(l[0], l[1], l[2])
    raise ValueError("Expected slices of length 1 to 3.")


def remove_duplicates(l):
    """Order-preserving de-duplication that also works for unhashable elements (dicts)."""
    out = []
    for elt in l:
        if elt not in out:
            out.append(elt)
    return out


def merge_dicts(*args):
    merged = {}
    for d in args:
        for k, v in d.items():
            if k in merged:
                raise Exception("can only merge dictionaries with unique keys")
            merged[k] = v
    return merged


def chunk(l, n):
    """Yield successive n-sized chunks from l."""
    if n == 0:
        return
    for i in range(0, len(l), n):
        yield l[i:i + n]
