/* oracle/linkin_127.c -- TEST INFRASTRUCTURE (oracle/Makefile.ref, target `linkin`): the reference's 127-mer build calls
 * call_pregraph (standardPregraph/main.c:72-75,341); the library keeps that behaviour under call_pregraph_127mer. */
int call_pregraph_127mer(int argc, char **argv);
int call_pregraph(int argc, char **argv) { return call_pregraph_127mer(argc, argv); }
