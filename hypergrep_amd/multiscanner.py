#! /usr/bin/env python3
"""GNU-grep style front end over hypergrep_amd.grep(): many files, one pattern set, results replayed in file order.

Counterpart of the reference's hypergrep/multiscanner.py (same function names, arguments, output lines and exit
codes: parallel_grep :86-223, print_results :226-255, read_stdin :258-270, to_basic_regular_expressions :273-305,
to_gnu_regular_expressions :308-328, parse_args :331-548, main :551-606), written for this engine:

* the scan of each file is one `hyperscan()` call of the native library, which streams the file into HBM and hands
  files to the node's GPUs round-robin — the pool here only has to keep a few calls in flight, it does no matching;
* finished files are reported by the calling thread as their futures complete (no pool callbacks), a file that
  finishes early waits in `parked` until every file before it has been printed;
* `--mp` uses spawned (not forked) worker processes: a forked child must not inherit an initialised HIP runtime.
"""
from __future__ import annotations

import argparse
import concurrent.futures
import multiprocessing
import os
import re
import sys
import textwrap
from typing import Any, Generator, Iterable

import hypergrep_amd

_BRE_LITERALS = "+?(){}|"  # regex operators in ERE / PCRE, ordinary characters in a POSIX basic regular expression


def _grep_with_index(index: int, args: Iterable, kwargs: dict[str, Any]) -> tuple[int, Any]:
    """One job: grep a file, hand back (job index, result or the exception it raised)."""
    try:
        return index, hypergrep_amd.grep(*args, **kwargs)
    except Exception as error:  # pylint: disable=broad-except
        return index, error


def get_argparse_files(args: argparse.Namespace) -> list[str]:
    """Files named on the command line.  As in GNU grep, once -e / -f supplies patterns the first positional is a file."""
    named = list(args.files or [])
    if args.pattern and (args.patterns or args.pattern_files):
        named.insert(0, args.pattern)
    return named


def get_argparse_patterns(args: argparse.Namespace) -> list[str]:
    """Patterns from -e, -f FILE (one per line) or, without either, the first positional.

    Raises ValueError (message in grep's style) for a pattern Python's `re` rejects or the engine cannot compile.
    """
    found: list[str] = list(args.patterns or [])
    if not found and not args.pattern_files and args.pattern:
        found.append(args.pattern)
    for name in args.pattern_files or []:
        with open(name, "rt", encoding="utf-8") as handle:
            found.extend(line.rstrip("\n") for line in handle.readlines())
    for pattern in found:  # cheap syntax check with a readable message before the engine's yes / no answer
        try:
            re.compile(pattern)
        except Exception as error:
            raise ValueError(f"hyperscanner: invalid regex: {error}") from error
    if hypergrep_amd.check_compatibility(found):
        raise ValueError(
            "hyperscanner: incompatible regex: for more information visit "
            "https://intel.github.io/hyperscan/dev-reference/compilation.html#unsupported-constructs"
        )
    return found


class _Replay:
    """Turns finished grep jobs into output lines, in file order when asked to, and keeps the exit-code facts."""

    def __init__(self, files: list, options: dict[str, Any]):
        self.files = files
        self.opt = options
        self.parked: dict[int, Any] = {}
        self.due = 0  # index of the next file to report in ordered mode
        self.total = 0
        self.matched = False
        self.errored = False

    def accept(self, index: int, outcome: Any) -> None:
        if self.opt["ordered_results"] and index != self.due:
            self.parked[index] = outcome
            return
        self._report(index, outcome)
        while self.due in self.parked:
            self._report(self.due, self.parked.pop(self.due))

    def _report(self, index: int, outcome: Any) -> None:
        name = self.files[index]
        opt = self.opt
        if isinstance(outcome, Exception):
            print(f"hyperscanner: {name}: {outcome}")  # grep's message shape
            self.errored = True
            self.due += 1
            return
        found, return_code = outcome
        if return_code:
            self.errored = True
        if found:
            self.matched = True
            if opt["quiet"]:
                return  # first match ends a quiet run; nothing is printed and the order no longer matters
        if opt["files_without_match"]:
            if not found:
                print(name)
        elif opt["files_with_matches"]:
            if found:
                print(name)
        elif opt["total_results"]:
            self.total += found
        elif opt["count_results"]:
            print(f"{name}:{found}" if opt["with_file_name"] else f"{found}")
        else:
            try:
                print_results(found, name, with_file_name=opt["with_file_name"], with_line_number=opt["with_line_number"])
            except BrokenPipeError:
                pass  # `| head` closed stdout: keep draining jobs quietly
        self.due += 1


def parallel_grep(  # pylint: disable=too-many-arguments,too-many-locals
    files: list,
    patterns: list[str],
    ignore_case: bool = False,
    ordered_results: bool = True,
    count_results: bool = False,
    total_results: bool = False,
    with_file_name: bool = False,
    with_line_number: bool = False,
    use_multithreading: bool = True,
    only_matching: bool = False,
    no_messages: bool = False,
    max_match_count: int = 0,
    files_without_match: bool = False,
    files_with_matches: bool = False,
    quiet: bool = False,
) -> int:
    """Search files for the patterns and print what grep would print.

    Args:
        files: files to scan (plain, gzip or zstd).
        patterns: regular expressions the engine accepts.
        ignore_case: case-insensitive matching.
        ordered_results: print a file's results only after those of every earlier file.
        count_results: print the number of matching lines per file instead of the lines.
        total_results: print one number, the matching lines of all files together.
        with_file_name / with_line_number: output prefixes.
        use_multithreading: threads (default) or, if False, spawned worker processes.
        only_matching: print each matched part on its own line.
        no_messages: no error text for missing / unreadable files.
        max_match_count: stop reading a file after that many matching lines (0: no limit).
        files_without_match / files_with_matches: print file names only; reading stops at the first match.
        quiet: print nothing and stop everything at the first match.

    Returns:
        grep's exit code: 2 after any error, else 1 without a match, else 0.
    """
    if files_without_match or files_with_matches or quiet:
        max_match_count = 1  # these modes only ask "is there a match"
    replay = _Replay(
        files,
        {"ordered_results": ordered_results, "count_results": count_results, "total_results": total_results, "with_file_name": with_file_name,
         "with_line_number": with_line_number, "files_without_match": files_without_match, "files_with_matches": files_with_matches, "quiet": quiet},
    )
    job_kwargs = {
        "ignore_case": ignore_case,
        "count_only": count_results or total_results,
        "only_matching": only_matching,
        "no_messages": no_messages,
        "max_match_count": max_match_count,
    }
    # the reference runs one job per core; here a job is a GPU scan with its own reader threads and pinned buffers, so a
    # handful in flight already keeps every GPU of the node busy
    workers = max(1, min(len(files), max((os.cpu_count() or 2) - 1, 1), 16))
    if use_multithreading:
        pool: concurrent.futures.Executor = concurrent.futures.ThreadPoolExecutor(max_workers=workers)
    else:
        pool = concurrent.futures.ProcessPoolExecutor(max_workers=workers, mp_context=multiprocessing.get_context("spawn"))
    try:
        jobs = [pool.submit(_grep_with_index, index, (name, patterns), job_kwargs) for index, name in enumerate(files)]
        for done in concurrent.futures.as_completed(jobs):
            replay.accept(*done.result())
            if quiet and replay.matched:
                for job in jobs:
                    job.cancel()
                break
    finally:
        pool.shutdown(wait=True, cancel_futures=True)
    if total_results:
        print(replay.total)
    return 2 if replay.errored else (0 if replay.matched else 1)


def print_results(results: list, file_name: str, with_file_name: bool = False, with_line_number: bool = False) -> None:
    """Print (line number, line) results with the requested prefixes; lines keep their own newline."""
    head = f"{file_name}:" if with_file_name else ""
    if with_line_number:
        chunks = [f"{head}{number}:{line}" for number, line in results]
    else:
        chunks = [f"{head}{line}" for _number, line in results]
    if chunks:
        print("".join(chunks), end="")


def read_stdin() -> Generator[str, None, None]:
    """File names from standard input, one per line, up to the first empty line or end of input."""
    for raw in iter(sys.stdin.readline, ""):
        name = raw.strip()
        if not name:
            return
        yield name


def to_basic_regular_expressions(patterns: list[str]) -> list[str]:
    """Read the patterns as POSIX basic regular expressions and return their ERE / PCRE spelling.

    In a BRE the characters + ? ( ) { } | are literals and become operators when escaped, the opposite of ERE / PCRE:
    every one of them swaps its escaping.  Raises ValueError when the result no longer compiles.
    """
    converted = []
    for pattern in patterns:
        out: list[str] = []
        for position, char in enumerate(pattern):
            if char in _BRE_LITERALS:
                if position and pattern[position - 1] == "\\":
                    out[-1] = char  # escaped in the BRE: an operator, spelled bare
                else:
                    out.append("\\" + char)  # bare in the BRE: a literal, spelled escaped
            else:
                out.append(char)
        text = "".join(out)
        try:
            re.compile(text)
        except Exception as error:
            raise ValueError(f"hyperscanner: invalid regex: {error}") from error
        converted.append(text)
    return converted


def to_gnu_regular_expressions(patterns: list[str]) -> list[str]:
    """GNU's word-edge operators \\< and \\> become \\b (not for PCRE input, which is taken as written).

    An operator whose backslash directly follows another backslash is left alone.
    """
    converted = []
    for pattern in patterns:
        out: list[str] = []
        position = 0
        while position < len(pattern):
            char = pattern[position]
            if char == "\\" and pattern[position + 1 : position + 2] in ("<", ">") and (position == 0 or pattern[position - 1] != "\\"):
                out.append("\\b")
                position += 2
            else:
                out.append(char)
                position += 1
        converted.append("".join(out))
    return converted


def parse_args(args: list = None) -> argparse.Namespace:
    """Command line of the `hyperscanner` command: grep's option letters where grep has them."""
    parser = argparse.ArgumentParser(
        formatter_class=argparse.RawTextHelpFormatter,
        add_help=False,  # -h is grep's --no-filename
        description=textwrap.dedent(
            """\
            grep over many files with one multi-pattern scan per file on the GPU.
              hyperscanner <regex> <file(s)>
              find <args> | hyperscanner <regex>
            Patterns are limited to what the engine compiles (no look-around, no back-references); plain, gzip and
            zstd files are read; only the options listed here exist (nothing is passed on to a grep process)."""
        ),
    )
    parser.add_argument("pattern", nargs="?", help="Regex pattern to use.")
    parser.add_argument("files", nargs="*", help="Files to scan.")
    parser.add_argument_group("Generic Program Information").add_argument(
        "--help", action="help", default=argparse.SUPPRESS, help="show this help message and exit"
    )

    syntax = parser.add_argument_group("Pattern Syntax").add_mutually_exclusive_group()
    syntax.set_defaults(regexp="bre")
    syntax.add_argument("-E", "--extended-regexp", dest="regexp", action="store_const", const="ere", help="PATTERNS are extended regular expressions.")
    syntax.add_argument("-G", "--basic-regexp", dest="regexp", action="store_const", const="bre", help="PATTERNS are basic regular expressions (default).")
    syntax.add_argument("-P", "--perl-regexp", dest="regexp", action="store_const", const="pcre", help="PATTERNS are Perl-compatible regular expressions.")

    matching = parser.add_argument_group("Matching Control")
    matching.add_argument("-e", "--regexp", action="append", dest="patterns", metavar="pattern", help="A pattern; may be repeated and combined with -f.")
    matching.add_argument("-f", "--file", action="append", dest="pattern_files", metavar="file", help="Read patterns from FILE, one per line; may be repeated.")
    matching.add_argument("-i", "--ignore-case", action="store_true", help="Case-insensitive matching.")

    output = parser.add_argument_group("General Output Control")
    output.add_argument("-c", "--count", action="store_true", help="Print the number of matching lines per file.")
    output.add_argument("-L", "--files-without-match", action="store_true", help="Print only the names of files without a match.")
    output.add_argument("-l", "--files-with-matches", action="store_true", help="Print only the names of files with a match.")
    output.add_argument("-m", "--max-count", type=int, default=0, help="Stop reading a file after NUM matching lines.")
    output.add_argument("-o", "--only-matching", action="store_true", help="Print only the matched parts, one per line.")
    output.add_argument("-q", "--quiet", "--silent", action="store_true", help="Print nothing; exit 0 at the first match.")
    output.add_argument("-s", "--no-messages", action="store_true", help="No messages about missing or unreadable files.")

    prefix = parser.add_argument_group("Output Line Prefix Control")
    names = prefix.add_mutually_exclusive_group()
    names.add_argument("-H", "--with-filename", action="store_true", default=None, help="Prefix lines with the file name (default with several files).")
    names.add_argument("-h", "--no-filename", action="store_true", default=None, help="No file name prefix (default with one file).")
    prefix.add_argument("-n", "--line-number", action="store_true", help="Prefix lines with their 1-based line number.")

    parser.add_argument_group("File and Directory Selection").add_argument(
        "-a", "--text", action="store_true", help="Accepted for grep compatibility; files are always read as bytes."
    )

    own = parser.add_argument_group("Unique arguments to hyperscanner")
    own.add_argument("-t", "--total", action="store_true", help="Print one count of matching lines over all files.")
    own.add_argument("--no-gnu", dest="gnu_regexp", action="store_false", help="Keep GNU operators such as \\< as written (BRE / ERE input only).")
    own.add_argument("--no-order", dest="ordered", action="store_false", help="Print each file's results as soon as it finishes.")
    own.add_argument("--no-sort", dest="sort_files", action="store_false", help="Keep the given file order instead of sorting the names.")
    own.add_argument("--mp", action="store_false", dest="use_multithreading", help="Worker processes instead of threads.")
    parser.set_defaults(parser=parser)
    return parser.parse_intermixed_args(args=args)


def main() -> None:
    """The `hyperscanner` command."""
    args = parse_args()
    try:
        patterns = get_argparse_patterns(args)
        if not patterns:
            args.parser.print_usage()
            raise SystemExit(2)
        if args.regexp == "bre":
            patterns = to_basic_regular_expressions(patterns)
    except ValueError as error:
        print(error)
        raise SystemExit(2) from error
    if args.gnu_regexp and args.regexp != "pcre":
        patterns = to_gnu_regular_expressions(patterns)
    files = get_argparse_files(args) or list(read_stdin())
    if args.sort_files:
        files = sorted(files)
    if not files:
        args.parser.print_usage()
        raise SystemExit(2)
    if args.no_filename is not None:
        with_file_name = False
    elif args.with_filename is not None:
        with_file_name = True
    else:
        with_file_name = len(files) > 1
    raise SystemExit(
        parallel_grep(
            files=files,
            patterns=patterns,
            ignore_case=args.ignore_case,
            ordered_results=args.ordered,
            count_results=args.count,
            total_results=args.total,
            with_file_name=with_file_name,
            with_line_number=args.line_number,
            use_multithreading=args.use_multithreading,
            only_matching=args.only_matching,
            no_messages=args.no_messages,
            max_match_count=args.max_count,
            quiet=args.quiet,
            files_without_match=args.files_without_match,
            files_with_matches=args.files_with_matches,
        )
    )


if __name__ == "__main__":
    try:
        main()
    except KeyboardInterrupt as user_interrupt:
        raise SystemExit(130) from user_interrupt
