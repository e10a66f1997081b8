#!/usr/bin/env python3
"""Generate tests/golden/*.json from the reference's own test fixtures.

Runs ONLY in the build container (reads /root/reference, which does not exist on the GPU box);
the emitted JSON files are committed so `pytest` never touches the reference at run time.
Sources (all under /root/reference):
  tests/flow/social/{person,countries,friends,visits}.csv + social_queries.py   (in-tree social graph
      and the expected result sets of its traversal queries)
  tests/flow/test_bfs.py:10-213            (5-node algo.BFS graph and known answers — transcribed below
      because they live inside test methods, cited line by line)
  tests/flow/test_expand_into.py:17-95     (ExpandInto counts 1 / 2 / 1 / 4)
  tests/flow/test_multiple_edges.py:11-96  (multi-edge lifecycle: ids 0 -> 1 after deleting the first)
  graph/src/graph/graphblas/versioned_matrix.rs:1278-1330, 1399-1523 (fold thresholds, LCG model test)
  graph/src/graph/graphblas/matrix.rs:1617-1695 (dup-collapse + grown coordinate sets)
"""
import csv
import importlib.util
import json
import os
import sys

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_social():
    d = os.path.join(REF, "tests/flow/social")
    rd = lambda f: list(csv.reader(open(os.path.join(d, f))))
    persons = [{"name": r[0], "age": int(r[1]), "gender": r[2], "status": r[3]} for r in rd("person.csv")]
    countries = [r[0] for r in rd("countries.csv")]
    friends = [[r[0], r[1]] for r in rd("friends.csv")]
    visits = [[r[0], r[1], r[2]] for r in rd("visits.csv")]
    # expected results straight from the reference's QueryInfo objects
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "tests/flow"))
    spec = importlib.util.spec_from_file_location("social_queries", os.path.join(d, "social_queries.py"))
    sq = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sq)
    want = ["my_friends_query", "friends_of_friends_query", "friends_of_friends_single_and_over_30_query",
            "friends_of_friends_visited_netherlands_and_single_query", "relation_type_counts"]
    queries = {}
    for w in want:
        q = getattr(sq, w)
        queries[w] = {"query": " ".join(q.query.split()), "expected": q.expected_result}
    return {"source": "tests/flow/social/*.csv + social_queries.py", "persons": persons, "countries": countries,
            "friends": friends, "visits": visits, "queries": queries}


def bfs5():
    # tests/flow/test_bfs.py:10-25: (a)-[:E1]->(b)-[:E1]->(c), (b)-[:E2]->(d)-[:E1]->(e)
    nodes = ["a", "b", "c", "d", "e"]
    edges = [["a", "b", "E1"], ["b", "c", "E1"], ["b", "d", "E2"], ["d", "e", "E1"]]  # edge ids 0..3
    cases = [
        # (source, max_depth, rel_type, expected nodes (unordered) | null = no row)   file:line
        {"src": "a", "depth": -1, "type": None, "nodes": ["b", "c", "d", "e"], "line": "test_bfs.py:63-76"},
        {"src": "a", "depth": -1, "type": "E1", "nodes": ["b", "c"], "line": "test_bfs.py:104-113"},
        {"src": "a", "depth": -1, "type": "E1", "nodes": ["b", "c"], "edges_dst": ["b", "c"], "line": "test_bfs.py:124-152"},
        {"src": "b", "depth": -1, "type": "E1", "nodes": ["c"], "edges_dst": ["c"], "line": "test_bfs.py:124-152"},
        {"src": "c", "depth": -1, "type": "E1", "nodes": None, "line": "test_bfs.py:124-136 (absent row)"},
        {"src": "d", "depth": -1, "type": "E1", "nodes": ["e"], "edges_dst": ["e"], "line": "test_bfs.py:124-152"},
        {"src": "e", "depth": -1, "type": "E1", "nodes": None, "line": "test_bfs.py:124-136 (absent row)"},
        {"src": "a", "depth": 1, "type": None, "nodes": ["b"], "edges_dst": ["b"], "line": "test_bfs.py:155-168"},
        {"src": "b", "depth": 1, "type": None, "nodes": ["c", "d"], "edges_dst": ["c", "d"], "line": "test_bfs.py:171-199"},
        {"src": "d", "depth": 1, "type": None, "nodes": ["e"], "edges_dst": ["e"], "line": "test_bfs.py:171-199"},
        {"src": "c", "depth": 1, "type": None, "nodes": None, "line": "test_bfs.py:171-183 (absent row)"},
        {"src": "a", "depth": -1, "type": "NONE_EXISTING_RELATION", "nodes": None, "line": "test_bfs.py:201-208"},
        {"src": "e", "depth": -1, "type": None, "nodes": None, "line": "test_bfs.py:210-215 (leaf)"},
    ]
    return {"source": "tests/flow/test_bfs.py", "nodes": nodes, "edges": edges, "cases": cases}


def expand_into():
    # tests/flow/test_expand_into.py:17-95
    return {"source": "tests/flow/test_expand_into.py", "cases": [
        {"name": "test01_single_hop_no_multi_edge", "line": "17-36", "nodes": 2, "labels": {"0": ["A"], "1": ["B"]},
         "edges": [[0, 1, "R"]], "a": 0, "b": 1, "types": ["R"], "count_named_edge": 1, "count_pairs": 1},
        {"name": "test02_single_hop_multi_edge", "line": "39-61", "nodes": 2, "labels": {"0": ["A"], "1": ["B"]},
         "edges": [[0, 1, "R"], [0, 1, "R"]], "a": 0, "b": 1, "types": ["R"], "count_named_edge": 2, "count_pairs": 1},
        {"name": "test03_multi_hop_no_multi_edge", "line": "64-78", "nodes": 3, "labels": {"0": ["A"], "2": ["B"]},
         "edges": [[0, 1, "R"], [1, 2, "R"]], "a": 0, "b": 2, "two_hop_rows": 1},
        # (a)-[:R]->(i)-[:R]->(b) twice: the anonymous 2-hop pattern yields ONE row per (a, b) pair; the
        # test's count of 4 = 2 x 2 variable-length trails of the first MATCH x that single row
        {"name": "test04_multi_hop_multi_edge", "line": "80-95", "nodes": 3, "labels": {"0": ["A"], "1": ["B"]},
         "edges": [[0, 2, "R"], [2, 1, "R"], [0, 2, "R"], [2, 1, "R"]], "a": 0, "b": 1, "two_hop_rows": 1,
         "varlen_trails": 4, "expected_count": 4},
    ]}


def multiple_edges():
    # tests/flow/test_multiple_edges.py:11-96: edge ids seen by `MATCH (a)-[e:R]->(b) RETURN ID(e)`
    return {"source": "tests/flow/test_multiple_edges.py:11-96", "steps": [
        {"op": "none", "ids": []},
        {"op": "add", "id": 0, "ids": [0]},
        {"op": "add", "id": 1, "ids": [0, 1]},
        {"op": "del", "id": 0, "ids": [1]},
        {"op": "del", "id": 1, "ids": []},
        {"op": "add", "id": 2, "ids": [2]},
    ]}


def rust_unit_pins():
    return {
        "source": "graph/src/graph/graphblas/{matrix,versioned_matrix}.rs #[test]s",
        "dup_collapse": {"line": "matrix.rs:1686-1695 (build_bool_tolerates_duplicate_pairs, literal vectors)", "dim": 8,
                         "rows": [1, 3, 1, 3, 1], "cols": [2, 4, 2, 4, 2], "nvals": 2, "present": [[1, 2], [3, 4]]},
        "fold_thresholds": {"line": "versioned_matrix.rs:1278-1330", "READ_FOLD_K": 82000, "WRITE_FOLD_K": 20500000,
                            "MIN_FOLD_DELTA": 256, "threshold_read_tx1": 287, "threshold_write_tx1": 4528,
                            "threshold_read_tx100": 2864, "ratio": 15},
        "lcg_model": {"line": "versioned_matrix.rs:1380-1472", "seed": 0x5EED1234, "mul": 6364136223846793005,
                      "add": 1442695040888963407, "shift": 33, "steps": 4000, "dim": 512, "check_stride": 37,
                      "key": "((r % 24) * 7, (r / 24 % 24) * 11)"},
        "refold_probe": {"line": "versioned_matrix.rs:1481-1523", "filler": 1024, "probe": [7, 11],
                         "trigger": [300, 301], "dim": 512, "final_nvals": 1026},
        "grown_coords": {"line": "matrix.rs:1617-1672", "n": 64, "a": "(i, (7 i) % 48)", "b": "(i, (11 i + 3) % 48)"},
    }


def load_imdb():
    """tests/flow/imdb: 284 movies + 1 300-odd actors from the csv files, loaded exactly as imdb_utils.py does
    (CREATE lists the actors first, then the movies: node ids follow that order; one `act` edge per csv row whose
    movie exists), and the expected results of the traversal-only queries of imdb_queries.py."""
    import types
    d = os.path.join(REF, "tests/flow/imdb")
    movies = [r[0] for r in csv.reader(open(os.path.join(d, "resources/movies.csv")))]
    movie_set = set(movies)
    actors, seen, edges = [], set(), []
    for r in csv.reader(open(os.path.join(d, "resources/actors.csv"))):
        name, movie = r[0], r[2]
        if name not in seen:
            seen.add(name)
            actors.append(name)
        if movie in movie_set:
            edges.append([name, movie])
    # imdb_queries.py does `from imdb import QueryInfo`: give it a stand-in that just records the fields
    stub = types.ModuleType("imdb")

    class QueryInfo:
        def __init__(self, query=None, description=None, expected_result=None, reversible=True, **kw):
            self.query, self.description, self.expected_result = query, description, expected_result
    stub.QueryInfo = QueryInfo
    sys.modules["imdb"] = stub
    spec = importlib.util.spec_from_file_location("imdb_queries", os.path.join(d, "imdb_queries.py"))
    iq = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(iq)
    q = iq.IMDBQueries()
    want = ["number_of_actors_query", "actors_played_with_nicolas_cage_query",
            "actors_played_in_movie_straight_outta_compton_query", "how_many_movies_cameron_diaz_played_query",
            "grand_budapest_hotel_cast_and_their_other_roles"]
    queries = {w: {"query": " ".join(getattr(q, w).query.split()), "expected": getattr(q, w).expected_result}
               for w in want}
    return {"source": "tests/flow/imdb/resources/{movies,actors}.csv + imdb_queries.py (imdb_utils.py load order)",
            "actors": actors, "movies": movies, "act": edges, "queries": queries}


def main():
    assert os.path.isdir(REF), "run in the build container (needs /root/reference)"
    for name, obj in [("social.json", load_social()), ("bfs5.json", bfs5()), ("expand_into.json", expand_into()),
                      ("multiple_edges.json", multiple_edges()), ("rust_unit_pins.json", rust_unit_pins()),
                      ("imdb.json", load_imdb())]:
        with open(os.path.join(OUT, name), "w") as f:
            json.dump(obj, f, indent=1, sort_keys=True)
        print("wrote", name)


if __name__ == "__main__":
    main()
