"""Model of the multi-wave region growing of k_lsd_grow_mw (csrc/line_kernels.hip): one MAIN wave walks the seeds in order and is the only
writer of the committed USED map C; HELPER waves run ahead, grow regions of later seeds speculatively (reading C, marking their own pixels in
the shared owner map T) and hand the results over; main accepts a result iff none of the pixels it EVER accepted (first growth or
refinement) is committed at its turn AND every pixel it assumed used (because somebody else held it) is, else grows the region itself.

Why that rule is exact: C only grows, and a helper that read C(x) = 0 where the sequential algorithm would see USED(x) = 1 differs from it only
if it then ACCEPTS x (a pixel that is tested and rejected leaves no trace) -- so "no accepted pixel is committed at my turn" is precisely the
condition under which the sequential algorithm grows the same region from the same seed; and a helper that skipped x as used is right iff
x is used at its turn.  What a helper does about other waves' claims (a region growing from an EARLIER seed or by the main wave: the helper
gives up when it is about to accept such a pixel -- or, with `park` > 0, first WAITS (bounded) for that claim to change and looks at the
pixel again: park / resume, built and measured in round 5, not shipped; from a LATER seed: overridden; a finished region that waits for its turn: judged the same
way (policy 0) or assumed used (policy 1)) only changes how much speculation is wasted: the two checks above alone decide what is committed.

The model runs the protocol on toy images with a toy order-dependent region_grow (running mean angle, refinement that un-marks and regrows with
a tighter tolerance, radius reduction) under random interleavings of the waves, and requires the sequence of committed regions and the final
USED map to equal the sequential run.  It checks the protocol, not the arithmetic (tests/test_gpu_line.py does that on the GPU)."""
import random

import numpy as np
import pytest

GROUP = 8          # seeds per group (64 on the device)
LOOKAHEAD = 6      # groups a helper may be ahead of main
ENTRIES = 3        # results a helper keeps per group before it leaves the rest of the group to main


def neighbours(p, W, H):
    x, y = p % W, p // W
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            nx, ny = x + dx, y + dy
            if 0 <= nx < W and 0 <= ny < H:
                yield ny * W + nx


class Abort(Exception):
    pass


def grow(img, seed, tol, is_used, mark, poison=None, tick=None, park=None):
    """toy region_grow: breadth first over the list, running mean `angle`, acceptance depends on the order of acceptances.
    park(q): generator, True when the claim that stood in the way of q has changed (the pixel is then looked at AGAIN, from the used test on:
    the device repeats the whole round of seven points, nothing of which had been written), False when the wait ran out (give up)"""
    W, H, ang = img
    reg, s = [seed], float(ang[seed])
    mark(seed)
    i = 0
    while i < len(reg):
        for q in neighbours(reg[i], W, H):
            while True:
                if tick:
                    yield from tick()
                if is_used(q):
                    break
                if abs(ang[q] - s / len(reg)) <= tol:
                    if poison and poison(q):
                        if park and (yield from park(q)):
                            continue
                        raise Abort()
                    reg.append(q); s += float(ang[q]); mark(q)
                break
        i += 1
    return reg


def process_seed(img, seed, is_used, mark, unmark, poison=None, tick=None, park=None):
    """grow -> (maybe) refine: un-mark everything, regrow tighter -> (maybe) reduce: drop far pixels.  Returns (final list, every pixel ever
    accepted, line or None)"""
    W, H, ang = img
    first = yield from grow(img, seed, 0.30, is_used, mark, poison, tick, park)
    ever = list(first)
    final = first
    if len(first) >= 3 and (np.ptp(ang[first]) > 0.35):          # "density too low": refine
        for q in first:
            unmark(q)
        final = yield from grow(img, seed, 0.12, is_used, mark, poison, tick, park)
        ever += final
        if len(final) > 4 and np.ptp(ang[final]) > 0.15:         # reduce_region_radius: keep the points near the seed
            sx, sy = seed % W, seed // W
            keep = [q for q in final if abs(q % W - sx) + abs(q // W - sy) <= 2]
            for q in final:
                if q not in keep:
                    unmark(q)
            final = keep
    line = (seed, len(final), float(np.sum(ang[final]))) if len(final) >= 3 else None
    return final, ever, line


def run_gen(g):
    try:
        while True:
            next(g)
    except StopIteration as e:
        return e.value


def sequential(img, order):
    used = set()
    out = []
    for seed in order:
        if seed in used:
            continue
        own = set()
        final, ever, line = run_gen(process_seed(img, seed, lambda q: q in used or q in own, own.add, own.discard))
        used.update(final)
        out.append((seed, tuple(final), line))
    return out, used


def concurrent(img, order, n_helpers, rng, policy=0, claim_noise=0.0, park=0, park_spins=40):
    C, T = set(), {}                     # T: pixel -> helper that marked it last (the tentative-owner nibbles of the device)
    cur_pos = [0] * n_helpers            # seed position of each helper's latest attempt
    n_groups = (len(order) + GROUP - 1) // GROUP
    owner = [None] * n_groups          # None / "main" / helper id
    progress = [0] * n_groups          # positions of the group the owner has dealt with (published after the entry, if any)
    entries = [[] for _ in range(n_groups)]
    state = {"main_group": 0, "next_group": 0, "done": False, "wasted": 0, "used_spec": 0, "parked": 0, "resumed": 0}
    out = []

    def tick():
        yield

    def scramble(q):
        # The device updates a 4-bit claim with an atomicAnd followed by an atomicOr (line_kernels.hip set_used_t / tent_retag / tent_release): two waves
        # claiming one pixel at once can leave the OR of their ids -- a PHANTOM owner (another helper, a finished region, main) -- or wipe the other's
        # claim.  Claims are advisory: whatever they say, main validates every speculative region against the committed map.  claim_noise makes the
        # model do the same damage at random: the result must not change.
        if claim_noise and rng.random() < claim_noise:
            if rng.random() < 0.3:
                T.pop(q, None)
            else:
                T[q] = (rng.choice(["main"] + list(range(n_helpers))), rng.choice(["growing", "pending"]))

    def helper(hid):
        pending = []        # groups whose T marks this helper still has to clear (after main is through with them)
        while not state["done"]:
            for g in [g for g in pending if state["main_group"] > g]:
                for e in entries[g]:
                    for q in e["ever"]:
                        if T.get(q) == (hid, "pending"):     # another wave may have taken the pixel over
                            del T[q]
                pending.remove(g)
            g = state["next_group"]
            if g >= n_groups:
                yield
                if not pending:
                    return
                continue
            if g >= state["main_group"] + LOOKAHEAD or len(pending) >= 2:
                yield
                continue
            if owner[g] is not None:            # claimed meanwhile (the device does this with one atomic)
                state["next_group"] = max(state["next_group"], g + 1)
                continue
            owner[g] = hid; state["next_group"] = g + 1
            pending.append(g)
            for k, seed in enumerate(order[g * GROUP:(g + 1) * GROUP]):
                yield
                if len(entries[g]) >= ENTRIES:
                    break                       # out of result slots: main does the rest of this group itself
                my_pos = g * GROUP + k
                ow = T.get(seed)        # (holder, state): holder = helper id or "main", state = "growing" | "pending"
                if seed in C or (ow is not None and (ow[0] == "main" or ow[0] == hid or cur_pos[ow[0]] < my_pos)):
                    progress[g] = k + 1     # committed, or held by a region that will most likely swallow it
                    continue
                cur_pos[hid] = my_pos
                own, marked, assumed = set(), [], []        # own: this wave's private map; T only carries CLAIMS, which others may overwrite

                def mark(q, own=own, marked=marked):
                    own.add(q); T[q] = (hid, "growing"); marked.append(q)       # a later seed's claim is simply overwritten
                    scramble(q)

                def unmark(q, own=own):
                    own.discard(q)
                    if T.get(q) == (hid, "growing"):
                        del T[q]

                def is_used(q, my_pos=my_pos, assumed=assumed, own=own):
                    if q in C or q in own:
                        return True
                    o = T.get(q)
                    if policy == 1 and o is not None and o[1] == "pending":     # a FINISHED region that waits for its turn: assumed used, checked at this seed's turn
                        assumed.append(q)
                        return True
                    return False

                def poison(q, my_pos=my_pos):       # about to accept q: yield to the claim of a region growing from an earlier seed (main's always is)
                    o = T.get(q)
                    if o is None or o == (hid, "growing") or (policy == 1 and o[1] == "pending"):
                        return False
                    return o[0] == "main" or (o[0] != hid and cur_pos[o[0]] < my_pos)       # policy 0: a finished region's claim is judged like a growing one's
                parks = [park]

                def park_on(q, parks=parks):
                    # the claim in the way is a GROWING region's (a finished one that waits for its turn may wait long: no park); the wait ends when the
                    # claim on q has changed or q is committed, and it is bounded (a claim can be stale or a phantom: claim_noise) -- then: give up
                    o = T.get(q)
                    if parks[0] <= 0 or o is None or o[1] != "growing":
                        return False
                    parks[0] -= 1
                    state["parked"] += 1
                    for _ in range(park_spins):
                        yield
                        if T.get(q) != o or q in C:
                            state["resumed"] += 1
                            return True
                    return False
                try:
                    if len(marked) > 10_000:
                        raise Abort()
                    final, ever, line = yield from process_seed(img, seed, is_used, mark, unmark, poison=poison, tick=tick, park=park_on if park else None)
                    for q in final:                                 # finished: "helper is growing this" -> "a finished region that waits for its turn"
                        if T.get(q) == (hid, "growing"):
                            T[q] = (hid, "pending")
                    entries[g].append(dict(pos=k, seed=seed, final=final, ever=ever, line=line, assumed=assumed))
                except Abort:
                    for q in marked:
                        unmark(q)
                progress[g] = k + 1
            progress[g] = GROUP

    def main():
        for g in range(n_groups):
            state["main_group"] = g
            if owner[g] is None:
                owner[g] = "main"
                state["next_group"] = max(state["next_group"], g + 1)
            seeds = order[g * GROUP:(g + 1) * GROUP]
            for k, seed in enumerate(seeds):
                yield
                if seed in C:
                    continue
                e = None
                if owner[g] != "main":
                    while progress[g] <= k and not (entries[g] and entries[g][-1]["pos"] >= k):
                        yield                   # the helper is still busy with (or before) this position
                    e = next((x for x in entries[g] if x["pos"] == k), None)
                if e is not None and not any(q in C for q in e["ever"]) and all(q in C for q in e["assumed"]):
                    final, line = e["final"], e["line"]
                    state["used_spec"] += 1
                else:
                    if e is not None:
                        state["wasted"] += 1
                    own = set()

                    def m_mark(q, own=own):
                        own.add(q); T[q] = ("main", "growing")      # the helpers see what main is growing; main itself ignores every claim
                        scramble(q)

                    def m_unmark(q, own=own):
                        own.discard(q)
                        if T.get(q) == ("main", "growing"):
                            del T[q]
                    final, ever, line = yield from process_seed(img, seed, lambda q: q in C or q in own, m_mark, m_unmark, tick=tick)
                    C.update(final)
                    for q in ever:
                        if T.get(q) == ("main", "growing"):
                            del T[q]
                C.update(final)
                out.append((seed, tuple(final), line))
        state["main_group"] = n_groups
        state["done"] = True

    waves = [main()] + [helper(h) for h in range(n_helpers)]
    alive = list(range(len(waves)))
    steps = 0
    while alive:
        w = rng.choice(alive) if rng.random() < 0.8 else alive[0]
        try:
            next(waves[w])
        except StopIteration:
            alive.remove(w)
        steps += 1
        assert steps < 5_000_000, "the protocol does not terminate"
    return out, C, state


def toy_image(seed, W=20, H=14):
    r = np.random.default_rng(seed)
    ang = r.uniform(0, 1, W * H)
    # a few coherent "edges": rows / columns / blocks of nearly equal angle, so that regions are long and seeds of one edge are neighbours in the order
    for _ in range(10):
        a = r.uniform(0, 1)
        if r.uniform() < 0.5:
            y, x0, x1 = int(r.integers(0, H)), int(r.integers(0, W // 2)), int(r.integers(W // 2, W))
            ang[y * W + x0:y * W + x1] = a + r.normal(0, 0.03, x1 - x0)
        else:
            x, y0, y1 = int(r.integers(0, W)), int(r.integers(0, H // 2)), int(r.integers(H // 2, H))
            ang[x + W * np.arange(y0, y1)] = a + r.normal(0, 0.05, y1 - y0)
    order = list(np.argsort(-ang, kind="stable"))          # "gradient bins descending": neighbours on an edge are close in the order
    return (W, H, ang), [int(p) for p in order]


@pytest.mark.parametrize("seed", range(12))
def test_speculative_protocol_equals_the_sequential_scan(seed):
    img, order = toy_image(seed)
    want, want_used = sequential(img, order)
    assert len(want) > 20
    for n_helpers in (1, 3, 7):
        for policy in (0, 1):      # what the helpers make of a finished region's claim (MwLayout.policy on the device)
            rng = random.Random(1000 * seed + 10 * n_helpers + policy)
            got, used, st = concurrent(img, order, n_helpers, rng, policy)
            assert got == want, (seed, n_helpers, policy)
            assert used == want_used
    assert st["used_spec"] > 0          # the helpers did contribute


@pytest.mark.parametrize("seed", range(6))
def test_protocol_is_exact_whatever_the_claim_nibbles_say(seed):
    """the claim map is written with two non-atomic steps on the device: phantom owners and lost claims may only waste speculation"""
    img, order = toy_image(seed)
    want, want_used = sequential(img, order)
    for n_helpers in (3, 7):
        for policy in (0, 1):
            for noise in (0.05, 0.5):
                rng = random.Random(77 * seed + 10 * n_helpers + policy)
                got, used, st = concurrent(img, order, n_helpers, rng, policy, claim_noise=noise)
                assert got == want and used == want_used, (seed, n_helpers, policy, noise)


@pytest.mark.parametrize("seed", range(12))
def test_park_and_resume_equals_the_sequential_scan(seed):
    """region_grow<MW> with parking (round 5's experimental kernel, profiles/r05_latency_path_series.patch -- measured exact and 3 % slower, not shipped:
    profiles/r05_latency_path.md): a helper that is about to accept a pixel of an earlier seed's GROWING region waits for that claim to change and looks
    again, instead of leaving the seed to main.  Only the amount of useful speculation may change -- and the waits end: a
    helper only ever waits for a strictly earlier position, main never waits inside a region, every wait is bounded."""
    img, order = toy_image(seed)
    want, want_used = sequential(img, order)
    resumed = 0
    for n_helpers in (1, 3, 7):
        for policy in (0, 1):
            for park, spins in ((1, 40), (4, 200), (4, 3)):      # (4, 3): the wait mostly runs out -> the old give-up
                rng = random.Random(4000 * seed + 10 * n_helpers + policy + 100 * park + spins)
                got, used, st = concurrent(img, order, n_helpers, rng, policy, park=park, park_spins=spins)
                assert got == want, (seed, n_helpers, policy, park, spins)
                assert used == want_used
                resumed += st["resumed"]
    assert resumed > 0          # the waits did end with a second look somewhere


@pytest.mark.parametrize("seed", range(4))
def test_park_and_resume_with_scrambled_claims(seed):
    img, order = toy_image(seed)
    want, want_used = sequential(img, order)
    for n_helpers in (3, 7):
        for noise in (0.05, 0.5):
            rng = random.Random(91 * seed + n_helpers)
            got, used, st = concurrent(img, order, n_helpers, rng, 0, claim_noise=noise, park=3, park_spins=60)
            assert got == want and used == want_used, (seed, n_helpers, noise)
