"""TEST INFRASTRUCTURE ONLY (like everything under oracle/).

Which index does ``Tensor.topk`` keep when the k-th and (k+1)-th values are EXACTLY equal?  On the CPU ATen's
``topk`` (aten/src/ATen/native/cpu/TopKImpl.h, not vendored in the reference; torch 2.10.0) fills a vector of
``(value, index)`` pairs and runs libstdc++'s ``std::partial_sort`` when ``k * 64 <= n`` and otherwise
``std::nth_element(k - 1)`` + ``std::sort`` of the first k-1, all with a comparator that looks at the VALUE only.
The kept SET is therefore an artefact of introselect's pivoting / the heap's shape, and so is the ORDER equal values are
returned in -- which matters once: the reference's ``knn`` (util/util.py:159) drops the entry ``topk`` returns FIRST, so
when the best value of a row is shared (copies of a point; a neighbour whose distance rounds to the point's own) WHICH of the
tied entries is dropped comes out of that sort.  The HIP kNN kernels replay both for rows with such a tie
(vcr-net_amd/csrc/knn.hip, knn_tiebreak_kernel / tiebreak_row).  This module restates the libstdc++ algorithms
(bits/stl_algo.h: __introselect, __unguarded_partition_pivot, __move_median_to_first, __insertion_sort;
__introsort_loop, __final_insertion_sort; bits/stl_heap.h: __make_heap, __adjust_heap, __push_heap, __pop_heap,
__heap_select, __sort_heap) in plain Python;
tests/test_oracle_topk_ties.py pins it against ``torch.topk`` itself on tie-heavy inputs."""
import math


def comp(a, b):            # largest=True comparator on (value, index) pairs: value only
    return a[0] > b[0]

def move_median_to_first(q, result, a, b, c):
    if comp(q[a], q[b]):
        if comp(q[b], q[c]): q[result], q[b] = q[b], q[result]
        elif comp(q[a], q[c]): q[result], q[c] = q[c], q[result]
        else: q[result], q[a] = q[a], q[result]
    elif comp(q[a], q[c]): q[result], q[a] = q[a], q[result]
    elif comp(q[b], q[c]): q[result], q[c] = q[c], q[result]
    else: q[result], q[b] = q[b], q[result]

def unguarded_partition(q, first, last, pivot):
    while True:
        while comp(q[first], q[pivot]): first += 1
        last -= 1
        while comp(q[pivot], q[last]): last -= 1
        if not (first < last): return first
        q[first], q[last] = q[last], q[first]
        first += 1

def unguarded_partition_pivot(q, first, last):
    mid = first + (last - first) // 2
    move_median_to_first(q, first, first + 1, mid, last - 1)
    return unguarded_partition(q, first + 1, last, first)

def insertion_sort(q, first, last):
    if first == last: return
    for i in range(first + 1, last):
        if comp(q[i], q[first]):
            val = q[i]
            q[first + 1:i + 1] = q[first:i]
            q[first] = val
        else:   # unguarded linear insert
            val = q[i]; j = i
            while comp(val, q[j - 1]):
                q[j] = q[j - 1]; j -= 1
            q[j] = val

def push_heap(q, first, hole, top, val):
    parent = (hole - 1) // 2
    while hole > top and comp(q[first + parent], val):
        q[first + hole] = q[first + parent]
        hole = parent
        parent = (hole - 1) // 2
    q[first + hole] = val

def adjust_heap(q, first, hole, length, val):
    top = hole
    child = hole
    while child < (length - 1) // 2:
        child = 2 * (child + 1)
        if comp(q[first + child], q[first + child - 1]): child -= 1
        q[first + hole] = q[first + child]
        hole = child
    if (length & 1) == 0 and child == (length - 2) // 2:
        child = 2 * (child + 1)
        q[first + hole] = q[first + child - 1]
        hole = child - 1
    push_heap(q, first, hole, top, val)

def make_heap(q, first, last):
    length = last - first
    if length < 2: return
    parent = (length - 2) // 2
    while True:
        val = q[first + parent]
        adjust_heap(q, first, parent, length, val)
        if parent == 0: return
        parent -= 1

def pop_heap(q, first, last, result):
    val = q[result]
    q[result] = q[first]
    adjust_heap(q, first, 0, last - first, val)

def heap_select(q, first, middle, last):
    make_heap(q, first, middle)
    for i in range(middle, last):
        if comp(q[i], q[first]): pop_heap(q, first, middle, i)

def introselect(q, first, nth, last, depth_limit):
    while last - first > 3:
        if depth_limit == 0:
            heap_select(q, first, nth + 1, last)
            q[first], q[nth] = q[nth], q[first]
            return
        depth_limit -= 1
        cut = unguarded_partition_pivot(q, first, last)
        if cut <= nth: first = cut
        else: last = cut
    insertion_sort(q, first, last)

def nth_element(q, nth):
    n = len(q)
    if n == 0 or nth == n: return
    introselect(q, 0, nth, n, 2 * int(math.floor(math.log2(n))))

def topk_set_emulated(values, k):
    n = len(values)
    q = [(float(v), j) for j, v in enumerate(values)]
    if k * 64 <= n:
        heap_select(q, 0, k, n)            # partial_sort = heap_select + sort_heap (the SET is fixed by heap_select)
    else:
        nth_element(q, k - 1)
    return sorted(j for _, j in q[:k])



def sort_heap(q, first, last):
    while last - first > 1:
        last -= 1
        pop_heap(q, first, last, last)

def unguarded_linear_insert(q, last):
    val = q[last]
    nxt = last - 1
    while comp(val, q[nxt]):
        q[last] = q[nxt]; last = nxt; nxt -= 1
    q[last] = val

def introsort_loop(q, first, last, depth_limit):
    while last - first > 16:
        if depth_limit == 0:                # __partial_sort(first, last, last)
            heap_select(q, first, last, last)
            sort_heap(q, first, last)
            return
        depth_limit -= 1
        cut = unguarded_partition_pivot(q, first, last)
        introsort_loop(q, cut, last, depth_limit)
        last = cut

def std_sort(q, first, last):
    if first == last: return
    introsort_loop(q, first, last, 2 * int(math.floor(math.log2(last - first))))
    if last - first > 16:                   # __final_insertion_sort
        insertion_sort(q, first, first + 16)
        for i in range(first + 16, last): unguarded_linear_insert(q, i)
    else:
        insertion_sort(q, first, last)

def topk_order_emulated(values, k):
    """The k indices in the ORDER Tensor.topk(sorted=True) returns them (ties included)."""
    n = len(values)
    q = [(float(v), j) for j, v in enumerate(values)]
    if k * 64 <= n:
        heap_select(q, 0, k, n)
        sort_heap(q, 0, k)
    else:
        nth_element(q, k - 1)
        std_sort(q, 0, k - 1)
    return [j for _, j in q[:k]]
