"""The parallel KD-tree build of csrc/knn_tree.hip replaces nanoflann's planeSplit loop (nanoflann.hpp:1016-1043, a Hoare
partition with an unsigned right pointer) by its closed form: with c elements satisfying the predicate, the i-th violator among
the first c positions (ascending) is swapped with the i-th satisfier among the rest (descending), and the pointers meet at c.
This test pins that closed form to a transcription of the loop on random slices full of ties (CPU only)."""
import random


def plane_split_loop(ind, val, cutval):
    ind = list(ind)
    count = len(ind)
    left, right = 0, count - 1
    while True:
        while left <= right and val[ind[left]] < cutval:
            left += 1
        while right and left <= right and val[ind[right]] >= cutval:
            right -= 1
        if left > right or not right:
            break
        ind[left], ind[right] = ind[right], ind[left]
        left += 1
        right -= 1
    lim1, right = left, count - 1
    while True:
        while left <= right and val[ind[left]] <= cutval:
            left += 1
        while right and left <= right and val[ind[right]] > cutval:
            right -= 1
        if left > right or not right:
            break
        ind[left], ind[right] = ind[right], ind[left]
        left += 1
        right -= 1
    return ind, lim1, left


def closed_form_pass(ind, pred, lo):
    ind = list(ind)
    sat = [pred(x) for x in ind[lo:]]
    cnt = sum(sat)
    violators = [lo + p for p in range(cnt) if not sat[p]]                       # ascending
    satisfiers = [lo + p for p in range(len(sat) - 1, cnt - 1, -1) if sat[p]]    # descending
    assert len(violators) == len(satisfiers)
    for a, b in zip(violators, satisfiers):
        ind[a], ind[b] = ind[b], ind[a]
    return ind, lo + cnt


def plane_split_closed_form(ind, val, cutval):
    ind, lim1 = closed_form_pass(ind, lambda x: val[x] < cutval, 0)
    ind, lim2 = closed_form_pass(ind, lambda x: val[x] <= cutval, lim1)
    return ind, lim1, lim2


def test_closed_form_equals_the_loop():
    rng = random.Random(1)
    for _ in range(60000):
        n = rng.randrange(1, 48)
        vals = [rng.randrange(0, rng.choice([1, 2, 3, 5, 50])) for _ in range(n)]
        ind = list(range(n))
        rng.shuffle(ind)
        mn, mx = min(vals), max(vals)
        cut = min(max(rng.choice(vals + [mn, mx, (mn + mx) / 2]), mn), mx)  # middleSplit_ clamps cutval into [min, max]
        assert plane_split_loop(ind, vals, cut) == plane_split_closed_form(ind, vals, cut)
