"""User arms whose compiled kernels (abr_control_amd/specialize.py) the test-suite uses.  `__graft_entry__.build()`
builds them into the in-tree cache (abr_control_amd/_compiled_arms, which travels to the GPU box like libabrk.so);
one hipcc run each, skipped when the cached plugin matches the current kernel headers."""
from abr_control_amd import _abi, specialize


def test_arms():
    from tests.synthetic_arms import make_arm

    three = dict(_abi.load_table("threejoint"))
    three["name"] = "threejoint_user"
    return {
        # the built-in threejoint's table as a user arm: the plugin's kernels are the built-in's, bit for bit
        "threejoint_user": three,
        # four joints, non-orthogonal fixed rotations, an offset EE: nothing a built-in arm has
        "synthetic4": make_arm(4, 204, True),
    }


def build_all(verbose=False):
    import threading

    abi = specialize.plugin_abi(from_sources=True)
    specialize.prune(specialize.IN_TREE, abi)  # plugins of older kernel headers would only travel to the GPU box for nothing
    out, err = {}, []

    def one(name, tab):
        try:
            out[name] = specialize.compile_arm(tab, cache_dir=specialize.IN_TREE, abi=abi, verbose=verbose)
        except Exception as e:  # noqa: BLE001 - re-raised on the caller's thread
            err.append(e)

    ths = [threading.Thread(target=one, args=kv) for kv in test_arms().items()]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    if err:
        raise err[0]
    return out
